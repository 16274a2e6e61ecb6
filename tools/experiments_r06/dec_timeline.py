#!/usr/bin/env python3
"""Timeline of the fused FFT decoder (gf_decode128_fft_plan_kernel) inside the Tx step from s_memrealtime stamps, by stamp set
(variant libraries: tools/experiments_r05/build_variant.sh stamps<set> -DFFT_STAMPS -DFFT_STAMP_SET=<set>; 0 = the kernel's phases,
1 = inside the plan, 2 = inside the size-64 inverse transform).
usage: SDRHIP_LIB_PATH=tools/experiments_r05/lib/libsdrhip_stamps<set>.so python tools/experiments_r06/dec_timeline.py <set> [tx_gather]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrdaemon_amd as sd
from sdrdaemon_amd import _lib
import signals
import headline_inputs as hi

SET = int(sys.argv[1]) if len(sys.argv) > 1 else 0
NAMES = {
    0: ["start", "plan done", "first 32 loads landed, their copy stores issued", "inverse64 + t5 fold", "exchange (2 barriers)", "forward16 + syndromes (2 barriers)", None, "Minv x syndromes + stores"],
    1: ["start", "headers + tables in LDS (barrier 1)", "block counts (LDS atomics, barrier 2)", "classification: ballots, ranks, x / y / rpos (barriers 3, 4)",
        "position + row maps written", "logarithm sums over the N x N pairs (barrier)", "Minv by table look-up (barrier) = plan done", "rest of the kernel"],
    2: ["start", "plan done", "first 32 loads landed, their copy stores issued", "blocks 0..30 of the transform (elements 0..31)", "mid: other 32 loads landed, their copy stores, parity",
        "blocks 31..62", "t5 fold", "rest of the kernel"],
}[SET]
ctx = sd.Context(0)
S, n = 8, 1 << 25
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
meta = {"tv_sec": 1, "tv_usec": 0, "center_frequency_khz": 435000, "sample_rate": 625000, "nb_fec": 32}
rxf, keep = hi.tx_received_frames(ctx, x, meta)
del x
ctx.set_option("dec_max_rows", 32)
if len(sys.argv) > 2:
    ctx.set_option("tx_gather", int(sys.argv[2]))
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
cols = [k for k in range(8) if NAMES[k] is not None]
print("stamp set %d: waves stamped %d, kernel span %.1f us" % (SET, len(st), (st[:, 7].max() - t0) * 0.01))
for k in cols:
    print("%-70s at p0 %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  p100 %6.2f us" % ((NAMES[k],) + tuple(np.percentile((st[:, k] - t0) * 0.01, [0, 10, 50, 90, 100]))))
for a, b in zip(cols[:-1], cols[1:]):
    dd = (st[:, b] - st[:, a]) * 0.01
    print("%-70s mean %6.2f us  p10 %6.2f  p90 %6.2f" % (NAMES[b], dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
