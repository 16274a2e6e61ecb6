#!/usr/bin/env python3
"""Timeline of the FFT encoder's waves inside the Rx step (8 x 2^25, decimate16 + framing + CM256 128+32) from s_memrealtime stamps:
variant library built with -DFFT_STAMPS.  usage: SDRHIP_LIB_PATH=tools/experiments_r05/lib/libsdrhip_fftstamps.so python tools/experiments_r05/fft_stamps.py"""
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
rx = sd.RxPipe(ctx, S, log2decim=4, nb_fec=32, pipelined=False)
for i in range(30):
    rx.process_view(x, i, 0)
torch.cuda.synchronize()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * (8192 * 8))()
lib.sdrhip_debug_fft_stamps.argtypes = [ctypes.c_void_p]
assert lib.sdrhip_debug_fft_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
st = np.concatenate([st, (np.arange(8192) & 3)[:, None]], axis=1)  # column 8: the wave's index in its workgroup (ch = w & 1, hf = w >> 1)
st = st[st[:, 0] > 0]
st = st[st[:, 0] > st[:, 0].max() - 100000]  # the last launch only (1 ms)
t0 = st[:, 0].min()
print("waves stamped:", len(st), " kernel span %.1f us" % ((st[:, 5].max() - t0) * 0.01))
for k, name in enumerate(["start", "tables in LDS", "64 loads landed, copy stores issued, parity", "inverse64 + t5 fold", "exchange (2 barriers, t6 + stage 4 in the hf = 1 wave)", "forward16 + 16 rows stored"]):
    print("%-62s at (us after the first start): p0 %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  p100 %6.2f" % ((name,) + tuple(np.percentile((st[:, k] - t0) * 0.01, [0, 10, 50, 90, 100]))))
for k, name in enumerate(["table fill", "loads", "inverse64 + fold", "exchange", "forward16 + rows"]):
    dd = (st[:, k + 1] - st[:, k]) * 0.01
    print("%-30s mean %6.2f us  p10 %6.2f  p90 %6.2f" % (name, dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
hw = st[:, 6]
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0x1) << 7)  # cu_id, se_id, sh_id (within the XCC)
late = st[:, 0] - t0 > 500
print("waves that started > 5 us after the first: %d; their starts (us): %s" % (late.sum(), np.sort(np.unique(((st[late, 0] - t0) * 0.01).round(0)))[:20]))

# which SIMD does wave w of a workgroup run on?  (hf = 1 waves carry 320 constant multiplications, hf = 0 waves 209)
simd = (hw >> 4) & 3
tab = np.zeros((4, 4), dtype=int)
for w, sd in zip(st[:, 8], simd):
    tab[int(w), int(sd)] += 1
print("wave index in the workgroup (rows) x SIMD (columns):")
print(tab)
for w in range(4):
    m = st[:, 8] == w
    print("wave %d (ch %d, hf %d): inverse64 + fold mean %.2f us, end of wave p50 %.2f p90 %.2f us" % (w, w & 1, w >> 1, ((st[m, 3] - st[m, 2]) * 0.01).mean(), np.percentile((st[m, 5] - t0) * 0.01, 50), np.percentile((st[m, 5] - t0) * 0.01, 90)))
