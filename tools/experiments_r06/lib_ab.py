#!/usr/bin/env python3
"""One line of step / kernel times (Rx step 8 x 2^25, Tx step 8 x 128 frames with random 24-erasure patterns) for the library that is
loaded (SDRHIP_LIB_PATH): the A / B partner of build variants, run interleaved from a shell loop.
usage: SDRHIP_LIB_PATH=... python tools/experiments_r06/lib_ab.py <label> [option=value ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import headline_inputs as hi  # noqa: E402
import sdrdaemon_amd as sd  # noqa: E402
import signals  # noqa: E402
from sdrdaemon_amd.engine import K_DECIMATE, K_FEC_DECODE, K_FEC_ENCODE, K_INTERPOLATE  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "lib"
ctx = sd.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v) if v.lstrip("-").isdigit() else v)
S, n = 8, 1 << 25
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
meta = {"tv_sec": 1, "tv_usec": 0, "center_frequency_khz": 435000, "sample_rate": 625000, "nb_fec": 32}
rxf, keep = hi.tx_received_frames(ctx, x, meta)
rx = sd.RxPipe(ctx, S, log2decim=4, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=32, center_frequency_khz=435000, sample_rate=625000)
tx = sd.TxPipe(ctx, S, 4)
ctx.set_option("dec_max_rows", 32)


def timed(fn, classes, steps=100):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    ctx.set_option("ktime_stride", 4)
    ctx.kernel_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    per = {}
    for c in classes:
        t, k = ctx.kernel_timing_read(c)
        per[c] = t / max(k, 1)
    ctx.kernel_timing(False)
    return ms, per


for r in range(2):
    ms, per = timed(lambda: rx.process_view(x, 1, 0), [K_DECIMATE, K_FEC_ENCODE])
    ms2, per2 = timed(lambda: tx.process(rxf), [K_FEC_DECODE, K_INTERPOLATE])
    print("%-28s Rx step %.4f ms  K1m %.4f  K3f %.4f   |   Tx step %.4f ms  decode %.4f  K5w %.4f" %
          (label, ms, per[K_DECIMATE], per[K_FEC_ENCODE], ms2, per2[K_FEC_DECODE], per2[K_INTERPOLATE]), flush=True)
