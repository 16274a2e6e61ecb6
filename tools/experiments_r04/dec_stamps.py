#!/usr/bin/env python3
"""Phase timeline of gf_decode128_kernel (K4s) from s_memrealtime stamps (100 MHz) written by thread 0 of every workgroup:
variant library built with -DDEC128_STAMPS (tools/experiments_r04/dec_stamps.patch).
usage: SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_stamps.so python tools/experiments_r04/dec_stamps.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
from sdrdaemon_amd import _lib  # noqa: E402

ctx = sd.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
F, R, Stx = 128, 32, 8
frames = torch.randint(0, 256, (Stx * F, 128, 512), generator=g, device=dev, dtype=torch.uint8)
frames[:, :, 2] = torch.arange(128, device=dev, dtype=torch.uint8)
rec = sd.fec_encode_frames(ctx, frames, R)
allb = torch.cat([frames, rec], dim=1)
rs = np.random.RandomState(3)
keep = np.stack([np.sort(rs.permutation(160)[:136])[:128] for _ in range(Stx * F)])
rx = allb[torch.arange(Stx * F, device=dev)[:, None], torch.from_numpy(keep).to(dev)].contiguous().reshape(Stx * F, 128, 512)
ctx.set_option("dec_max_rows", 32)
for _ in range(5):
    out = sd.fec_decode_frames(ctx, rx)
ctx.synchronize()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * (4096 * 8))()
lib.sdrhip_debug_dec_stamps.argtypes = [ctypes.c_void_p]
assert lib.sdrhip_debug_dec_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8)[:2 * Stx * F, :7].astype(np.int64)
t0 = st[:, 0].min()
names = ["prologue (tables, plan -> LDS)", "walk (2 column blocks: loads, copy-out, convolution)", "atomics into ysum + barrier",
         "syndromes (recovery loads, kmul) + barrier", "Minv x syndromes", "stores"]
d = np.diff(st, axis=1) * 10.0  # ns
print("workgroups %d; kernel span %.1f us (first start .. last end)" % (len(st), (st[:, 6].max() - t0) * 0.01))
print("start of workgroups, us after the first: quartiles", np.percentile((st[:, 0] - t0) * 0.01, [0, 25, 50, 75, 100]).round(1))
for k, n in enumerate(names):
    print("%-56s mean %7.2f us   p10 %7.2f  p90 %7.2f" % (n, d[:, k].mean() / 1e3, np.percentile(d[:, k], 10) / 1e3, np.percentile(d[:, k], 90) / 1e3))
print("%-56s mean %7.2f us" % ("whole workgroup", (st[:, 6] - st[:, 0]).mean() * 0.01))
# residency: workgroups per CU over time
lib.sdrhip_debug_dec_occupancy.restype = ctypes.c_int
print("hipOccupancyMaxActiveBlocksPerMultiprocessor(gf_decode128_kernel, 256 threads): %d" % lib.sdrhip_debug_dec_occupancy())
full = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8)[:2 * Stx * F].astype(np.int64)
hw = full[:, 7]
key = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 8) & 0xf)
cus = np.unique(key)
print("CUs seen: %d" % len(cus))
ts = np.arange(0, (st[:, 6].max() - t0), 200)  # every 2 us
conc = np.array([((st[:, 0] - t0 <= t) & (st[:, 6] - t0 > t)).sum() for t in ts])
print("resident workgroups chip-wide every 10 us:", [int(c) for c in conc[::5]])
per_cu_max = [max(((st[key == c, 0] - t0 <= t) & (st[key == c, 6] - t0 > t)).sum() for t in ts[::2]) for c in cus[:64]]
print("max resident workgroups on a CU (first 64 CUs):", np.bincount(per_cu_max))
first = np.sort(st[:, 0] - t0) * 0.01
print("start time of the k-th workgroup, us: k=256 %.1f  512 %.1f  768 %.1f  1024 %.1f  1280 %.1f  1536 %.1f" % tuple(first[[255, 511, 767, 1023, 1279, 1535]]))
