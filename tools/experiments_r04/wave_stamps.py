#!/usr/bin/env python3
"""Timeline of K5w's waves (8 x 2^21 inputs = 8192 waves; interpolate16 by default) from s_memrealtime stamps (100 MHz) written by lane 0 of every wave:
variant library built with -DW_STAMPS (tools/experiments_r04/wave_stamps.patch + wave_stamps_kernels.patch).
usage: SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_wstamps.so python tools/experiments_r04/wave_stamps.py [log2interp]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
from sdrdaemon_amd import _lib  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = sd.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
S, n = 8, 1 << 21  # 8192 waves (the stamp buffer's size) whatever the ratio
n_out = n << L
x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
out = torch.empty((S, n_out, 2), dtype=torch.int16, device=dev)
ctx.set_option("interp_path", "wave")
d = sd.Interpolators(ctx, S)
for _ in range(5):
    d.interpolate(L, x, out=out)
ctx.synchronize()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * (8192 * 12))()
lib.sdrhip_debug_w_stamps.argtypes = [ctypes.c_void_p]
assert lib.sdrhip_debug_w_stamps(buf) == 0
nw = min(8192, S * ((n + 2047) // 2048))
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 12)[:nw].astype(np.int64)
t0 = st[:, 0].min()
print("interpolate%d: waves %d; kernel span %.1f us" % (1 << L, nw, (st[:, 3].max() - t0) * 0.01))
print("start of waves, us after the first: deciles", np.percentile((st[:, 0] - t0) * 0.01, range(0, 101, 10)).round(1))
print("end of waves: deciles", np.percentile((st[:, 3] - t0) * 0.01, range(0, 101, 10)).round(1))
for k, name in enumerate(["cluster + state + warm-up samples landed, state in LDS", "warm-up (44 inputs through the cascade, no stores)", "the segment's pairs (stores)"]):
    dd = (st[:, k + 1] - st[:, k]) * 0.01
    print("%-60s mean %7.2f us   p10 %7.2f  p90 %7.2f" % (name, dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
print("%-60s mean %7.2f us" % ("whole wave", ((st[:, 3] - st[:, 0]) * 0.01).mean()))

# chip-wide progress: pairs of blocks (2 x 2048 x 4 B x 2^(L-4) of output each) completed per 10-us bin
pe = (st[:, 4:12] - t0).reshape(-1) * 0.01
bins = np.arange(0, pe.max() + 10, 10)
h, _ = np.histogram(pe, bins)
per_pair_bytes = 2 * 128 * (1 << L) * 4
print("time bin (us): stores completed in the bin as TB/s")
print("  ".join("%d:%.1f" % (b, c * per_pair_bytes / 10e-6 / 1e12) for b, c in zip(bins[:-1], h)))
