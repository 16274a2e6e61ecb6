#!/usr/bin/env python3
"""decimate16_cen on the matrix cores (8 x 2^25): LDS-DMA ring depth 4 against depth 3 (interleaved rounds), and the span
lengths around 32 768 (the 0.975-ms cliff of profiles/r04_decim_paths.txt).  usage: python tools/bench_ring.py [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
from sdrdaemon_amd.engine import K_DECIMATE  # noqa: E402

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S, n, L = 8, 1 << 25, 4
ctx = sd.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
out = torch.empty((S, n >> L, 2), dtype=torch.int16, device=dev)


def timed(fn, reps=40, preroll_s=0.25):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < preroll_s:
        fn()
        ctx.synchronize()
    ctx.kernel_timing(True)
    for _ in range(reps):
        fn()
    ms, cnt = ctx.kernel_timing_read(K_DECIMATE)
    ctx.kernel_timing(False)
    return ms / max(cnt, 1)


ctx.set_option("decim_path", "mfma")
ref = None
res = {}
for r in range(ROUNDS):
    for ring in [int(v) for v in os.environ.get("RINGS", "4,3").split(",")]:
        ctx.set_option("mfma_ring", ring)
        ctx.set_option("mfma_span", 0)
        d = sd.Decimators(ctx, S, 0)
        ms = timed(lambda: d.decimate(L, 2, 16, x, out=out))
        d2 = sd.Decimators(ctx, S, 0)
        y, _ = d2.decimate(L, 2, 16, x)
        ctx.synchronize()
        if ref is None:
            ref = y.clone()
        assert torch.equal(ref, y), ("ring", ring)
        res.setdefault(ring, []).append(ms)
for ring, v in res.items():
    print("ring depth %d: %s ms per launch (bit-identical outputs)" % (ring, " ".join("%.4f" % t for t in v)), flush=True)
if os.environ.get("NOSWEEP"):
    sys.exit(0)
for ring in (4, 3):
    ctx.set_option("mfma_ring", ring)
    for span in (30720, 31744, 32768, 33792, 34816, 36864):
        ctx.set_option("mfma_span", span)
        d = sd.Decimators(ctx, S, 0)
        ms = timed(lambda: d.decimate(L, 2, 16, x, out=out), reps=15)
        p = d.last_plan()
        print("ring %d span %6d: %.4f ms  (waves per stream %d, pieces %d, tail_start %d)" % (ring, span, ms, p["wps"], p["npieces"], p["tail_start"]), flush=True)
