#!/usr/bin/env python3
"""VALU vs matrix-core centred decimator kernels on the GPU box (hipEvent kernel-class timers).
usage: python tools/bench_decim_paths.py [log2 samples per stream, default 25] [streams, default 8]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
from sdrdaemon_amd.engine import K_DECIMATE  # noqa: E402

ONLY = None
if len(sys.argv) > 1 and ":" in sys.argv[1]:  # path:span:log2decim, e.g. mfma:0:4 (for profiling one kernel)
    ONLY = sys.argv.pop(1).split(":")
LOGN = int(sys.argv[1]) if len(sys.argv) > 1 else 25
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = sd.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
n = 1 << LOGN
x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)


def timed(fn, reps=int(os.environ.get("REPS", "30")), preroll_s=0.25):
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


for L in (6, 5, 4, 3, 2) if ONLY is None else (int(ONLY[2]),):
    out = torch.empty((S, n >> L, 2), dtype=torch.int16, device=dev)
    ref = None
    for path, span in ((("valu", 0), ("mfma", 0), ("mfma", 8192), ("mfma", 16384), ("mfma", 32768), ("mfma", 65536)) if ONLY is None
                       else ((ONLY[0], int(ONLY[1])),)):
        ctx.set_option("decim_path", path)
        ctx.set_option("mfma_span", span)
        d = sd.Decimators(ctx, S, 0)
        ms = timed(lambda: d.decimate(L, 2, 16, x, out=out))
        d2 = sd.Decimators(ctx, S, 0)
        y, _ = d2.decimate(L, 2, 16, x)
        ctx.synchronize()
        if ref is None:
            ref = y.clone()
        same = bool(torch.equal(ref, y))
        gs = S * n / ms / 1e6
        print("decimate%-2d_cen %s span %6d: %7.4f ms  %7.1f Gsamples/s  %6.0f GB/s = %.1f %% of 8 TB/s  (== valu: %s)" %
              (1 << L, path, span, ms, gs, gs * (4 + 4 / (1 << L)), gs * (4 + 4 / (1 << L)) / 80, same), flush=True)
