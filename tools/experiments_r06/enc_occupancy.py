#!/usr/bin/env python3
"""Where do the FFT encoder's workgroups of one Rx step run?  From the -DFFT_STAMPS library (s_memrealtime + HW_ID per wave): workgroups per CU,
the ones that start late and what ran on their CU before them.
usage: SDRHIP_LIB_PATH=tools/experiments_r05/lib/libsdrhip_fftstamps.so python tools/experiments_r06/enc_occupancy.py [frames standalone]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrdaemon_amd as sd
from sdrdaemon_amd import _lib
import signals

ctx = sd.Context(0)
lib = _lib.lib()
lib.sdrhip_debug_fft_stamps.argtypes = [ctypes.c_void_p]

def read():
    buf = (ctypes.c_ulonglong * (8192 * 8))()
    assert lib.sdrhip_debug_fft_stamps(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
    st = np.concatenate([st, (np.arange(8192) >> 2)[:, None]], axis=1)  # column 8: workgroup
    st = st[st[:, 0] > 0]
    return st[st[:, 0] > st[:, 0].max() - 100000]

def report(st, what):
    t0 = st[:, 0].min()
    hw = st[:, 6]
    cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0x1) << 7)
    wg = st[:, 8]
    xcd = wg & 7  # (workgroups are dealt round robin over the XCDs)
    key = xcd * 1024 + cu
    start = (st[:, 0] - t0) * 0.01
    end = (st[:, 5] - t0) * 0.01
    w0 = (np.arange(len(st)) % 4) == 0  # one row per workgroup (its wave 0) where all four are stamped
    first = {}
    for k, s, e, g in zip(key, start, end, wg):
        first.setdefault(k, []).append((s, e, g))
    per_cu = {k: len({g for _, _, g in v}) for k, v in first.items()}
    hist = np.bincount(list(per_cu.values()))
    print("%s: %d waves, %d workgroups on %d CUs; workgroups per CU: %s; launch %.1f us" % (what, len(st), len(set(wg)), len(per_cu), {i: int(n) for i, n in enumerate(hist) if n}, end.max()))
    late = sorted({(g, k) for k, s, g in zip(key, start, wg) if s > 5.0})
    print("  workgroups that started > 5 us after the first: %d" % len(late))
    for g, k in late[:24]:
        v = first[k]
        mine = [x for x in v if x[2] == g]
        others = sorted({(round(min(s for s, _, gg in v if gg == o), 1), round(max(e for _, e, gg in v if gg == o), 1)) for o in {x[2] for x in v} if o != g})
        print("    wg %4d on xcd %d cu 0x%02x: start %.1f end %.1f; the other %d workgroups of that CU (start, end): %s" % (g, k >> 10, k & 1023, min(s for s, _, _ in mine), max(e for _, e, _ in mine), len(others), others))
    by_n = {}
    for k, n in per_cu.items():
        by_n.setdefault(n, []).append(max(e for _, e, _ in first[k]))
    for n in sorted(by_n):
        print("  CUs with %d workgroups: last wave ends p50 %.1f  max %.1f us" % (n, np.percentile(by_n[n], 50), max(by_n[n])))

if len(sys.argv) > 1:
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    for F in [int(a) for a in sys.argv[1:]]:
        fr = torch.randint(0, 256, (F, 128, 512), generator=g, device=dev, dtype=torch.uint8)
        fr[:, :, 2] = torch.arange(128, device=dev, dtype=torch.uint8)
        for _ in range(20):
            sd.fec_encode_frames(ctx, fr, 32)
        torch.cuda.synchronize()
        report(read(), "stand-alone, %d frames" % F)
else:
    S, n = 8, 1 << 25
    x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
    rx = sd.RxPipe(ctx, S, log2decim=4, nb_fec=32, pipelined=False)
    for i in range(30):
        rx.process_view(x, i, 0)
    torch.cuda.synchronize()
    report(read(), "Rx step, 8 x 2^25")
