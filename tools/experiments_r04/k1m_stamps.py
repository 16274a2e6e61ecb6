#!/usr/bin/env python3
"""Timeline of K1m's waves (decimate16_cen, 8 x 2^25) from s_memrealtime stamps (100 MHz) + HW_ID / XCC_ID of every wave:
variant library built with -DMF_STAMPS (tools/experiments_r04/k1m_stamps.patch).
usage: SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_mfstamps.so python tools/experiments_r04/k1m_stamps.py [log2decim]"""
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
S, n = 8, 1 << 25
x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
d = sd.Decimators(ctx, S)
for _ in range(int(os.environ.get("ITERS", "300"))):
    y, _ = d.decimate(L, 2, 16, x)
ctx.synchronize()
plan = d.last_plan()
print("plan:", plan)
lib = _lib.lib()
buf = (ctypes.c_ulonglong * (8192 * 4))()
lib.sdrhip_debug_mf_stamps.argtypes = [ctypes.c_void_p]
assert lib.sdrhip_debug_mf_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4).astype(np.int64)
mf = st[:4096]
mf = mf[mf[:, 0] > 0]
va = st[4096:]
va = va[va[:, 0] > 0]
t0 = min(mf[:, 0].min(), va[:, 0].min())
end = max(mf[:, 1].max(), va[:, 1].max())
print("matrix-core waves %d, VALU piece waves %d; launch span %.1f us" % (len(mf), len(va), (end - t0) * 0.01))
print("matrix-core waves: start deciles (us)", np.percentile((mf[:, 0] - t0) * 0.01, range(0, 101, 10)).round(1))
print("matrix-core waves: end deciles (us)  ", np.percentile((mf[:, 1] - t0) * 0.01, range(0, 101, 10)).round(1))
dur = (mf[:, 1] - mf[:, 0]) * 0.01
print("matrix-core waves: duration mean %.1f  min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us" % (dur.mean(), dur.min(), *np.percentile(dur, [10, 50, 90]), dur.max()))
print("VALU pieces: start deciles", np.percentile((va[:, 0] - t0) * 0.01, range(0, 101, 10)).round(1))
print("VALU pieces: end deciles  ", np.percentile((va[:, 1] - t0) * 0.01, range(0, 101, 10)).round(1))
xcc = (mf[:, 2] >> 32) & 0xf
hw = mf[:, 2] & 0xffffffff
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
print("per XCC: waves, mean duration, mean end")
for xid in sorted(set(xcc)):
    m = xcc == xid
    print("  xcc %d: %4d waves  duration %.1f us  end %.1f us (max %.1f)" % (xid, m.sum(), dur[m].mean(), ((mf[m, 1] - t0) * 0.01).mean(), ((mf[m, 1] - t0) * 0.01).max()))
key = xcc * 1000 + se * 100 + cu
u, cnt = np.unique(key, return_counts=True)
print("CUs used by matrix-core waves: %d; waves per CU: min %d max %d" % (len(u), cnt.min(), cnt.max()))
slow = np.argsort(-dur)[:8]
print("slowest waves (duration, xcc, se, cu, simd):", [(round(float(dur[i]), 1), int(xcc[i]), int(se[i]), int(cu[i]), int(simd[i])) for i in slow])
print("VALU piece workgroups (stream, piece): start .. end us, xcc / se / cu of wave 0")
raw = st[4096:4096 + 4 * 64].reshape(-1, 4, 4)
npieces = plan["npieces"]
for lx in range(S * npieces):
    w0 = raw[lx, 0]
    if w0[0] <= 0:
        continue
    print("  (%d, %d): %7.1f .. %7.1f   xcc %d se %d cu %d" % (lx // npieces, lx % npieces, (w0[0] - t0) * 0.01, (raw[lx, :, 1].max() - t0) * 0.01,
                                                             (w0[2] >> 32) & 0xf, (w0[2] >> 13) & 7, (w0[2] >> 8) & 0xf))
