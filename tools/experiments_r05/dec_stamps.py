#!/usr/bin/env python3
"""Timeline of the FFT decoder's waves inside the Tx step (bench.py's configs[3]: 8 x 128 frames, a distinct 24-erasure pattern per frame) from
s_memrealtime stamps: variant library built with -DFFT_STAMPS.
usage: SDRHIP_LIB_PATH=tools/experiments_r05/lib/libsdrhip_fftstamps.so python tools/experiments_r05/dec_stamps.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrdaemon_amd as sd
from sdrdaemon_amd import _lib
import signals
import headline_inputs as hi

ctx = sd.Context(0)
S, n = 8, 1 << 25
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
meta = {"tv_sec": 1, "tv_usec": 0, "center_frequency_khz": 435000, "sample_rate": 625000, "nb_fec": 32}
rxf, keep = hi.tx_received_frames(ctx, x, meta)
del x
ctx.set_option("dec_max_rows", 32)
tx = sd.TxPipe(ctx, S, hi.TX_LOG2_INTERP)
for i in range(30):
    tx.process(rxf)
torch.cuda.synchronize(); ctx.synchronize()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * (8192 * 8))()
lib.sdrhip_debug_fft_stamps.argtypes = [ctypes.c_void_p]
assert lib.sdrhip_debug_fft_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
st = st[st[:, 0] > 0]
st = st[st[:, 0] > st[:, 0].max() - 100000]
t0 = st[:, 0].min()
print("waves stamped:", len(st), " kernel span %.1f us" % ((st[:, 7].max() - t0) * 0.01))
cols = [0, 1, 2, 3, 4, 5, 7]
names = ["start", "tables + plan in LDS", "64 loads landed, copy stores issued, parity", "inverse64 + t5 fold", "exchange (2 barriers)", "forward16 + syndromes (2 barriers)", "Minv x syndromes + stores"]
for k, name in zip(cols, names):
    print("%-48s at (us after the first start): p0 %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  p100 %6.2f" % ((name,) + tuple(np.percentile((st[:, k] - t0) * 0.01, [0, 10, 50, 90, 100]))))
for (a, b), name in zip(zip(cols[:-1], cols[1:]), names[1:]):
    dd = (st[:, b] - st[:, a]) * 0.01
    print("%-48s mean %6.2f us  p10 %6.2f  p90 %6.2f" % (name, dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
