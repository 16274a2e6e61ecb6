#!/usr/bin/env python3
"""K1m launch time against launch size (round 6): decimate16_cen alone on banks of S streams x 2^k samples -- is there a fixed part per
launch?  (tools/bench_streams.sh in the r06 evidence set: 16 / 32 streams x 2^25 run 0.2144 / 0.2156 ms per 2^28 samples, 8 streams
0.2335.)  usage: python tools/experiments_r06/k1m_size_scan.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
from sdrdaemon_amd.engine import K_DECIMATE  # noqa: E402

ctx = sd.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
big = torch.randint(-32768, 32768, (1 << 31,), generator=g, device=dev, dtype=torch.int16)  # 4 GiB = 2^30 samples


def run(S, k, reps=24):
    n = 1 << k
    x = big[:S * n * 2].view(S, n, 2)
    out = torch.empty((S, n >> 4, 2), dtype=torch.int16, device=dev)
    d = sd.Decimators(ctx, S)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        d.decimate(4, 2, 16, x, out=out)
        ctx.synchronize()
    ctx.set_option("ktime_stride", 1)
    ctx.kernel_timing(True)
    for _ in range(reps):
        d.decimate(4, 2, 16, x, out=out)
    ms, cnt = ctx.kernel_timing_read(K_DECIMATE)
    ctx.kernel_timing(False)
    p = d.last_plan()
    per = ms / cnt
    print("%2d streams x 2^%d: %.4f ms per launch = %.4f ms per 2^28 samples  (%.1f %% of HBM peak)   plan: %s waves/stream x 8 spans of %d, path %s" %
          (S, k, per, per * (1 << 28) / (S * n), 4.25 * S * n / (per * 1e-3) / 8e12 * 100, p["wps"], p["span"], p["path"]), flush=True)
    del out, d


for rnd in range(2):
    for S, k in ((8, 23), (8, 24), (8, 25), (8, 26), (8, 27), (16, 24), (16, 25), (16, 26), (32, 24), (32, 25), (4, 26), (4, 27), (2, 27), (2, 28), (1, 27), (1, 28), (1, 29), (64, 24)):
        run(S, k)
