#!/usr/bin/env python3
"""Is the launch time of the matrix-core decimator a function of WHERE its buffers lie?  (fast / slow 'box states')"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdrdaemon_amd as sd
from sdrdaemon_amd.engine import K_DECIMATE

ctx = sd.Context(0)
dev = torch.device("cuda", 0)
S, n, L = 8, 1 << 25, 4
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
d = sd.Decimators(ctx, S, 0)


def timed(out, reps=40):
    for _ in range(60):
        d.decimate(L, 2, 16, x, out=out)
    ctx.synchronize()
    ctx.kernel_timing(True)
    for _ in range(reps):
        d.decimate(L, 2, 16, x, out=out)
    ms, cnt = ctx.kernel_timing_read(K_DECIMATE)
    ctx.kernel_timing(False)
    return ms / max(cnt, 1)


keep = []
print("x at %#x" % x.data_ptr())
for k in range(6):
    out = torch.empty((S, n >> L, 2), dtype=torch.int16, device=dev)
    keep.append(out)
    print("fresh out %d at %#x: %.4f ms" % (k, out.data_ptr(), timed(out)), flush=True)
    keep.append(torch.empty((3 << 20) * (k + 1), dtype=torch.uint8, device=dev))  # shift the next allocation
pool = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
per = (n >> L) * 4
for off in (0, 4096, 65536, 1 << 20, (1 << 21) + 8448, (5 << 20) + 256, 33 << 20):
    stride = per + 8192  # rows 16-byte aligned, not a power of two apart
    v = torch.as_strided(pool[off:].view(torch.int16), (S, n >> L, 2), (stride // 2, 2, 1))
    print("pool offset %#10x stride %#x: %.4f ms" % (off, stride, timed(v)), flush=True)
for rep in range(3):
    print("again fresh 0: %.4f ms   fresh 3: %.4f ms" % (timed(keep[0]), timed(keep[6])), flush=True)
