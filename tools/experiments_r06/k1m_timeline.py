#!/usr/bin/env python3
"""Timeline of K1m's matrix-core waves inside the Rx step (8 x 2^25, decimate16_cen, frame-layout stores) from s_memrealtime stamps
(variant library: tools/experiments_r05/build_variant.sh mfstamps -DMF_STAMPS): prologue, ring fill, pace by quarter of the launch, end.
usage: SDRHIP_LIB_PATH=tools/experiments_r05/lib/libsdrhip_mfstamps.so python tools/experiments_r06/k1m_timeline.py [decimate-only]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrdaemon_amd as sd
from sdrdaemon_amd import _lib
import signals

ctx = sd.Context(0)
S, n = 8, 1 << 25
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
if len(sys.argv) > 1:
    d = sd.Decimators(ctx, S)
    for _ in range(30):
        d.decimate(4, 2, 16, x)
    plan = d.last_plan()
else:
    rx = sd.RxPipe(ctx, S, log2decim=4, nb_fec=32, pipelined=False)
    for i in range(30):
        rx.process_view(x, i, 0)
    plan = rx.last_plan()
torch.cuda.synchronize(); ctx.synchronize()
print("plan:", plan)
lib = _lib.lib()
buf = (ctypes.c_ulonglong * (4096 * 10))()
lib.sdrhip_debug_mf_stamps.argtypes = [ctypes.c_void_p]
assert lib.sdrhip_debug_mf_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 10).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
us = lambda a: a * 0.01
nper = (int(plan["span"]) + 1024) // 1024
print("waves %d; periods per wave %d (warm-up 1); launch span %.1f us" % (len(st), nper, us(st[:, 7].max() - t0)))
names = ["start", "constants loaded, 24 DMAs issued", "first group landed, first LDS read back", "period 0 (warm-up) done", "25 % of the periods", "50 %", "75 %", "end"]
for k, nm in enumerate(names):
    print("%-44s at p0 %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  p100 %7.2f us" % ((nm,) + tuple(np.percentile(us(st[:, k] - t0), [0, 10, 50, 90, 100]))))
q = [1, nper // 4, nper // 2, 3 * nper // 4, nper]
for a, b, pa, pb in ((3, 4, q[0], q[1]), (4, 5, q[1], q[2]), (5, 6, q[2], q[3]), (6, 7, q[3], q[4])):
    dt = us(st[:, b] - st[:, a]) / max(pb - pa, 1)
    print("periods %2d .. %2d: %.3f us per period (p10 %.3f, p90 %.3f) = %.4f us per step" % (pa, pb, dt.mean(), np.percentile(dt, 10), np.percentile(dt, 90), dt.mean() / 32))
print("prologue (start -> DMAs issued) mean %.2f us; ring fill (-> first group landed) mean %.2f us; warm-up period mean %.2f us" %
      (us(st[:, 1] - st[:, 0]).mean(), us(st[:, 2] - st[:, 1]).mean(), us(st[:, 3] - st[:, 2]).mean()))
xcc = (st[:, 8] >> 32) & 0xf
for xid in sorted(set(xcc.tolist())):
    m = xcc == xid
    print("  xcc %d: %4d waves  duration %.1f us  end mean %.1f max %.1f us" % (xid, m.sum(), us(st[m, 7] - st[m, 0]).mean(), us(st[m, 7] - t0).mean(), us(st[m, 7] - t0).max()))
