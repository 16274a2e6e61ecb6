#!/usr/bin/env python3
"""The drop-in mode as sdrdaemonrx sees it: one TestSource-sized block (65 536 samples) per synchronous call through host
pointers (SDRHIP_MEM_HOST): microseconds per call, against the same call on device-resident data."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402

ctx = sd.Context(0)
for n in (65536, 262144, 1 << 20):
    x = np.random.default_rng(1).integers(-32768, 32768, (n, 2), dtype=np.int16)
    xd = torch.from_numpy(x).cuda()
    for L in (4,):
        d = sd.Decimators(ctx, 1, 0)
        for _ in range(50):
            d.decimate(L, 2, 16, x)
        t0 = time.perf_counter()
        K = 400
        for _ in range(K):
            d.decimate(L, 2, 16, x)
        th = (time.perf_counter() - t0) / K * 1e6
        out = torch.empty((n >> L, 2), dtype=torch.int16, device="cuda")
        for _ in range(50):
            d.decimate(L, 2, 16, xd, out=out)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            d.decimate(L, 2, 16, xd, out=out)
            ctx.synchronize()
        td = (time.perf_counter() - t0) / K * 1e6
        print("decimate%d_cen, %7d samples per call: host pointers %7.1f us (%6.1f M samples/s)   device pointers + sync %7.1f us" % (1 << L, n, th, n / th, td), flush=True)

# the Rx pipe (decimate16_cen + framing + CM256 128+32) fed from host memory, one stream
for n in (65536, 262144, 1 << 20, 1 << 22, 1 << 24):
    x = np.random.default_rng(2).integers(-32768, 32768, (n, 2), dtype=np.int16)
    for pipelined in (False, True):
        rx = sd.RxPipe(ctx, 1, log2decim=4, nb_fec=32, pipelined=pipelined)
        K = max(8, min(400, (1 << 26) // n))
        for _ in range(min(K, 20)):
            rx.process(x, 1, 2)
        t0 = time.perf_counter()
        for _ in range(K):
            rx.process(x, 1, 2)
        th = (time.perf_counter() - t0) / K * 1e6
        print("rx pipe, %8d samples per call, host pointers%s: %8.1f us per call = %7.1f M samples/s" % (n, " (pipelined)" if pipelined else "", th, n / th), flush=True)
