#!/usr/bin/env python3
"""decimate16_cen on small and medium calls: VALU kernel vs matrix-core kernel (wall time per call, device-resident)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
from sdrdaemon_amd.engine import K_DECIMATE  # noqa: E402

ctx = sd.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for S, logn in ((1, 20), (1, 21), (1, 22), (1, 23), (1, 24), (1, 25), (8, 20), (8, 22)):
    n = 1 << logn
    x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
    out = torch.empty((S, n >> L, 2), dtype=torch.int16, device=dev)
    res = []
    for path in ("valu", "mfma"):
        ctx.set_option("decim_path", path)
        d = sd.Decimators(ctx, S, 0)
        for _ in range(20):
            d.decimate(L, 2, 16, x, out=out)
        ctx.synchronize()
        ctx.kernel_timing(True)
        for _ in range(50):
            d.decimate(L, 2, 16, x, out=out)
        ms, cnt = ctx.kernel_timing_read(K_DECIMATE)
        ctx.kernel_timing(False)
        res.append(ms / max(cnt, 1) * 1e3)
    print("decimate%d_cen %d x 2^%d samples: valu %8.1f us   mfma %8.1f us" % (1 << L, S, logn, res[0], res[1]), flush=True)
