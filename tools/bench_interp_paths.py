#!/usr/bin/env python3
"""Interpolator kernels on the GPU box (hipEvent kernel-class timers): K5 (valu), K5w (wave), K5m (mfma, when built in).
usage: python tools/bench_interp_paths.py [path:span:log2interp] [log2 outputs per stream, default 25] [streams, default 8]
       PATHS=valu:0,wave:0,wave:1024 LS=4,5 python tools/bench_interp_paths.py ..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
from sdrdaemon_amd.engine import K_INTERPOLATE  # noqa: E402

ONLY = None
if len(sys.argv) > 1 and ":" in sys.argv[1]:
    ONLY = sys.argv.pop(1).split(":")
LOGN = int(sys.argv[1]) if len(sys.argv) > 1 else 25
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = sd.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)


def timed(fn, reps=int(os.environ.get("REPS", "30")), preroll_s=0.25):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < preroll_s:
        fn()
        ctx.synchronize()
    ctx.kernel_timing(True)
    for _ in range(reps):
        fn()
    ms, cnt = ctx.kernel_timing_read(K_INTERPOLATE)
    ctx.kernel_timing(False)
    return ms / max(cnt, 1)


PATHS = [(a.split(":")[0], int(a.split(":")[1])) for a in os.environ.get("PATHS", "valu:0,wave:0").split(",")]
LS = [int(v) for v in os.environ.get("LS", "4,3,2").split(",")]
for L in LS if ONLY is None else (int(ONLY[2]),):
    n_out = 1 << LOGN
    n = n_out >> L
    x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
    out = torch.empty((S, n_out, 2), dtype=torch.int16, device=dev)
    ref = None
    for path, span in (PATHS if ONLY is None else ((ONLY[0], int(ONLY[1])),)):
        ctx.set_option("interp_path", path)
        ctx.set_option("interp_span", span)
        d = sd.Interpolators(ctx, S)
        if os.environ.get("PRECOPY"):
            # the input as a kernel in front of the interpolator leaves it (the Tx pipe's decoder writes the payload right before):
            # freshly WRITTEN, i.e. possibly still in the Infinity Cache / L2 when the interpolator reads it
            x2 = torch.empty_like(x)

            def step():
                x2.copy_(x)
                d.interpolate(L, x2, out=out)
            ms = timed(step)
        else:
            ms = timed(lambda: d.interpolate(L, x, out=out))
        y = sd.Interpolators(ctx, S).interpolate(L, x)
        ctx.synchronize()
        if ref is None:
            ref = y.clone()
        same = bool(torch.equal(ref, y))
        gs = S * n_out / ms / 1e6
        bps = 4 + 4 / (1 << L)
        print("interpolate%-2d_cen %s span %6d: %7.4f ms  %7.1f Gsamples/s out  %6.0f GB/s = %.1f %% of 8 TB/s  (== first: %s)" %
              (1 << L, path, span, ms, gs, gs * bps, gs * bps / 80, same), flush=True)
