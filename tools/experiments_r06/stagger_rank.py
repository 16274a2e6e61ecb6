#!/usr/bin/env python3
"""Staggered start with the phase taken from the workgroup's ARRIVAL RANK on its CU (fec_stagger_mod = 100 + m: an atomic counter per CU,
so that the co-resident workgroups of every CU are in different phases whatever the dispatcher's dealing is): the Tx step's decoder (K4f,
m = 4) and the Rx step's encoder (K3f, m = 5) at a sweep of the sleep per phase; with a -DFFT_STAMPS library (SDRHIP_LIB_PATH) also the
decoder's timeline by phase.  usage: python tools/experiments_r06/stagger_rank.py [rounds]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import headline_inputs as hi  # noqa: E402
import sdrdaemon_amd as sd  # noqa: E402
import signals  # noqa: E402
from sdrdaemon_amd import _lib  # noqa: E402
from sdrdaemon_amd.engine import K_DECIMATE, K_FEC_DECODE, K_FEC_ENCODE, K_INTERPOLATE  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = sd.Context(0)
S, n = 8, 1 << 25
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
meta = {"tv_sec": 1, "tv_usec": 0, "center_frequency_khz": 435000, "sample_rate": 625000, "nb_fec": 32}
rxf, keep = hi.tx_received_frames(ctx, x, meta)
rx = sd.RxPipe(ctx, S, log2decim=4, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=32, center_frequency_khz=435000, sample_rate=625000)
tx = sd.TxPipe(ctx, S, 4)
ctx.set_option("dec_max_rows", 32)
lib = _lib.lib()
have_stamps = hasattr(lib, "sdrhip_debug_fft_stamps")


def timed(fn, classes, steps=60):
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


def stamps():
    buf = (ctypes.c_ulonglong * (8192 * 8))()
    lib.sdrhip_debug_fft_stamps.argtypes = [ctypes.c_void_p]
    assert lib.sdrhip_debug_fft_stamps(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
    st = st[st[:, 0] > 0]
    st = st[st[:, 0] > st[:, 0].max() - 100000]
    return st


def timeline(tag):
    for _ in range(10):
        tx.process(rxf)
    torch.cuda.synchronize(); ctx.synchronize()
    st = stamps()
    t0 = st[:, 0].min()
    hw = st[:, 6]
    phase = (hw >> 40) & 0xff
    cu = ((hw >> 8) & 0xff) | (((hw >> 32) & 0xf) << 8)
    print("---- %s: waves stamped %d, kernel span %.1f us, CUs %d" % (tag, len(st), (st[:, 7].max() - t0) * 0.01, len(np.unique(cu))))
    # phases per CU
    per_cu = {}
    for c, p, b in zip(cu, phase, np.arange(len(st))):
        per_cu.setdefault(int(c), set()).add(int(p))
    hist = np.bincount([len(v) for v in per_cu.values()], minlength=6)
    wg_per_cu = np.bincount(np.unique(cu, return_counts=True)[1] // 4, minlength=8)
    print("distinct phases per CU (count of CUs with k phases, k = 0..): %s; workgroups per CU histogram: %s" % (hist.tolist(), wg_per_cu.tolist()))
    cols = [0, 1, 2, 3, 4, 5, 7]
    names = ["start", "plan (+ sleep) done", "64 loads landed, copy stores issued", "inverse64 + t5 fold", "exchange", "forward16 + syndromes", "Minv x syndromes + stores"]
    for ph in sorted(set(phase.tolist())):
        m = phase == ph
        print("phase %d (%d waves):" % (ph, m.sum()))
        for k, name in zip(cols, names):
            print("   %-40s at p0 %6.2f  p50 %6.2f  p100 %6.2f us" % ((name,) + tuple(np.percentile((st[m, k] - t0) * 0.01, [0, 50, 100]))))


if have_stamps:
    for mod, v in ((0, 0), (104, 4), (104, 8), (104, 12)):
        ctx.set_option("fec_stagger_mod", mod)
        ctx.set_option("fec_stagger", v)
        timeline("fec_stagger_mod %d fec_stagger %d" % (mod, v))
else:
    for r in range(rounds):
        for mod_d, mod_e, v in ((0, 0, 0), (104, 105, 2), (104, 105, 4), (104, 105, 6), (104, 105, 8), (104, 105, 12), (104, 105, 16), (4, 5, 8), (103, 104, 8)):
            ctx.set_option("fec_stagger", v)
            ctx.set_option("fec_stagger_mod", mod_e)
            ms, per = timed(lambda: rx.process_view(x, 1, 0), [K_DECIMATE, K_FEC_ENCODE])
            ctx.set_option("fec_stagger_mod", mod_d)
            ms2, per2 = timed(lambda: tx.process(rxf), [K_FEC_DECODE, K_INTERPOLATE])
            print("round %d  mod (enc %3d, dec %3d) fec_stagger %2d   Rx step %.4f ms  K1m %.4f  K3f %.4f   |   Tx step %.4f ms  decode %.4f  K5w %.4f" %
                  (r, mod_e, mod_d, v, ms, per[K_DECIMATE], per[K_FEC_ENCODE], ms2, per2[K_FEC_DECODE], per2[K_INTERPOLATE]), flush=True)
ctx.set_option("fec_stagger", 0)
ctx.set_option("fec_stagger_mod", 0)
