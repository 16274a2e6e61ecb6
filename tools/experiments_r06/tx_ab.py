#!/usr/bin/env python3
"""A / B of one context option on the Tx step (8 x 128 frames, random 24-erasure patterns, dec_max_rows = 32) and the Rx step (8 x 2^25),
interleaved rounds.  usage: python tools/experiments_r06/tx_ab.py <option> <value> [<value> ...] [--rounds N] [--rx]"""
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

args = sys.argv[1:]
rounds, do_rx = 3, False
if "--rounds" in args:
    i = args.index("--rounds")
    rounds = int(args[i + 1])
    del args[i:i + 2]
if "--rx" in args:
    args.remove("--rx")
    do_rx = True
opt, values = args[0], args[1:]
ctx = sd.Context(0)
S, n = 8, 1 << 25
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
meta = {"tv_sec": 1, "tv_usec": 0, "center_frequency_khz": 435000, "sample_rate": 625000, "nb_fec": 32}
rxf, keep = hi.tx_received_frames(ctx, x, meta)
rx = sd.RxPipe(ctx, S, log2decim=4, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=32, center_frequency_khz=435000, sample_rate=625000)
if not do_rx:
    del x, rx
ctx.set_option("dec_max_rows", 32)


def timed(fn, classes, steps=80):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
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


for r in range(rounds):
    for v in values:
        ctx.set_option(opt, v)
        tx = sd.TxPipe(ctx, S, 4)
        ms2, per2 = timed(lambda: tx.process(rxf), [K_FEC_DECODE, K_INTERPOLATE])
        line = "round %d  %s = %-10s Tx step %.4f ms  decode %.4f  K5w %.4f" % (r, opt, v, ms2, per2[K_FEC_DECODE], per2[K_INTERPOLATE])
        if do_rx:
            ms, per = timed(lambda: rx.process_view(x, 1, 0), [K_DECIMATE, K_FEC_ENCODE])
            line += "   |   Rx step %.4f ms  K1m %.4f  K3f %.4f" % (ms, per[K_DECIMATE], per[K_FEC_ENCODE])
        print(line, flush=True)
        del tx
