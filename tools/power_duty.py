#!/usr/bin/env python3
"""Is a kernel's launch time a function of the DUTY CYCLE it runs at?  (GPU box.)  The same launches back to back, and with the
host idling between them (synchronise + sleep): a kernel that runs at the board's power cap gets faster when the chip rests in
between -- its clock floats with the average power -- while a kernel bound by its own critical path does not care.
usage: python tools/power_duty.py [decim|decim32|decim64|one27|interp|interp2|enc]..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
from sdrdaemon_amd.engine import K_DECIMATE, K_FEC_ENCODE, K_INTERPOLATE  # noqa: E402

what = sys.argv[1:] or ["decim", "interp", "enc"]
ctx = sd.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)


def run(name, fn, kclass, gaps=(0.0, 0.0002, 0.0005, 0.001, 0.0)):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        fn()
        ctx.synchronize()
    for gap in gaps:
        # settle at this duty cycle first (power management averages over milliseconds), then measure
        for phase in (0, 1):
            if phase == 1:
                ctx.kernel_timing(True)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < (0.6 if phase == 0 else 0.4):
                fn()
                if gap > 0:
                    ctx.synchronize()
                    t1 = time.perf_counter()
                    while time.perf_counter() - t1 < gap:
                        pass
                n += 1
            ctx.synchronize()
            wall = time.perf_counter() - t0
        ms, cnt = ctx.kernel_timing_read(kclass)
        ctx.kernel_timing(False)
        k = ms / max(cnt, 1)
        print("%-28s host gap %6.0f us: kernel %7.4f ms   duty %3.0f %%" % (name, gap * 1e6, k, 100.0 * k * 1e-3 * n / wall), flush=True)


if "decim" in what:
    S, n = 8, 1 << 25
    x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
    out = torch.empty((S, n >> 4, 2), dtype=torch.int16, device=dev)
    d = sd.Decimators(ctx, S)
    run("decimate16_cen (K1m)", lambda: d.decimate(4, 2, 16, x, out=out), K_DECIMATE)
    del x, out
for L in (5, 6):  # decimate32 / 64_cen on the matrix cores (at the same power wall as decimate16?)
    if "decim%d" % (1 << L) in what:
        S, n = 8, 1 << 25
        x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
        out = torch.empty((S, n >> L, 2), dtype=torch.int16, device=dev)
        d = sd.Decimators(ctx, S)
        run("decimate%d_cen (K1m)" % (1 << L), lambda: d.decimate(L, 2, 16, x, out=out), K_DECIMATE)
        assert d.last_plan()["path"] == "mfma"
        del x, out, d
if "one27" in what:  # configs[2] literally: one stream of 2^27
    x = torch.randint(-32768, 32768, (1, 1 << 27, 2), generator=g, device=dev, dtype=torch.int16)
    out = torch.empty((1, 1 << 23, 2), dtype=torch.int16, device=dev)
    d = sd.Decimators(ctx, 1)
    run("decimate16_cen, 1 x 2^27 (K1m)", lambda: d.decimate(4, 2, 16, x, out=out), K_DECIMATE)
    del x, out, d
if "interp2" in what:  # interpolate2_cen: a single stage, the K5 kernel
    S, n_out = 8, 1 << 25
    x = torch.randint(-32768, 32768, (S, n_out >> 1, 2), generator=g, device=dev, dtype=torch.int16)
    out = torch.empty((S, n_out, 2), dtype=torch.int16, device=dev)
    u = sd.Interpolators(ctx, S)
    run("interpolate2_cen (K5)", lambda: u.interpolate(1, x, out=out), K_INTERPOLATE)
    del x, out, u
if "interp" in what:
    S, n_out = 8, 1 << 25
    x = torch.randint(-32768, 32768, (S, n_out >> 4, 2), generator=g, device=dev, dtype=torch.int16)
    out = torch.empty((S, n_out, 2), dtype=torch.int16, device=dev)
    u = sd.Interpolators(ctx, S)
    run("interpolate16_cen (K5w)", lambda: u.interpolate(4, x, out=out), K_INTERPOLATE)
    del x, out
if "enc" in what:
    F = 1040
    frames = torch.randint(0, 256, (F, 128, 512), generator=g, device=dev, dtype=torch.uint8)
    run("CM256 encode, 1040 frames", lambda: sd.fec_encode_frames(ctx, frames, 32), K_FEC_ENCODE)
