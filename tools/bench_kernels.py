#!/usr/bin/env python3
"""Per-kernel timings on the GPU box (hipEvent kernel-class timers of libsdrhip.so).
usage: python tools/bench_kernels.py [decim] [interp] [fec]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
from sdrdaemon_amd.engine import K_DECIMATE, K_FEC_DECODE, K_FEC_ENCODE, K_INTERPOLATE  # noqa: E402

what = sys.argv[1:] or ["decim", "interp", "fec"]
ctx = sd.Context(0)
if os.environ.get("SDRHIP_DEC_MAX"):
    ctx.set_option("dec_max_rows", os.environ["SDRHIP_DEC_MAX"])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
S = 8


def timed(cls, fn, reps=20, preroll_s=0.15):
    import time

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < preroll_s:  # clocks ramp over the first ~60 ms of load
        fn()
        ctx.synchronize()
    ctx.kernel_timing(True)
    for _ in range(reps):
        fn()
    ms, n = ctx.kernel_timing_read(cls)
    ctx.kernel_timing(False)
    return ms / max(n, 1)


if "decim" in what:
    n = 1 << 24
    x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
    for fc, name in ((2, "cen"), (0, "inf"), (1, "sup")):
        for L in range(1, 7):
            if fc != 2 and L < 3:
                continue
            d = sd.Decimators(ctx, S, 0)
            out = torch.empty((S, n >> L, 2), dtype=torch.int16, device=dev)
            ms = timed(K_DECIMATE, lambda: d.decimate(L, fc, 16, x, out=out))
            gs = S * n / ms / 1e6
            print("decimate%-2d_%s  %8.3f ms  %8.1f Gsamples/s in  %7.1f GB/s" % (1 << L, name, ms, gs, gs * (4 + 4 / (1 << L))))
    del x
if "interp" in what:
    for L in range(1, 7):
        n = (1 << 26) >> L
        x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
        u = sd.Interpolators(ctx, S)
        out = torch.empty((S, n << L, 2), dtype=torch.int16, device=dev)
        ms = timed(K_INTERPOLATE, lambda: u.interpolate(L, x, out=out))
        go = S * (n << L) / ms / 1e6
        print("interpolate%-2d_cen  %8.3f ms  %8.1f Gsamples/s out  %7.1f GB/s" % (1 << L, ms, go, go * (4 + 4 / (1 << L))))
        del x, out
if "fec" in what:
    F = int(os.environ.get("FEC_F", "2048"))
    frames = torch.randint(0, 256, (F, 128, 512), generator=g, device=dev, dtype=torch.uint8)
    frames[:, :, 2] = torch.arange(128, device=dev, dtype=torch.uint8)
    ms = timed(K_FEC_ENCODE, lambda: sd.fec_encode_frames(ctx, frames, 32))
    print("fec_encode 128+32  %8.3f ms / %d frames  %8.2f Mframes/s  %7.1f Gsamples/s-equivalent(decim16)" % (ms, F, F / ms / 1e3, F * 258064 / ms / 1e6))
    rec = sd.fec_encode_frames(ctx, frames, 32)
    allb = torch.cat([frames, rec], dim=1)
    keep = [i for i in range(160) if i not in set(range(1, 121, 5))][:128]
    rx = allb[:, keep].contiguous()
    idx = rx[:, :, 2].cpu().numpy()
    ms = timed(K_FEC_DECODE, lambda: sd.fec_decode_frames(ctx, rx, idx))
    print("fec_decode 24 erasures (one pattern)  %8.3f ms / %d frames  %8.2f Mframes/s" % (ms, F, F / ms / 1e3))
if "tx" in what:
    import time

    F, R = 128, 32
    Stx = 8
    frames = torch.randint(0, 256, (Stx * F, 128, 512), generator=g, device=dev, dtype=torch.uint8)
    frames[:, :, 2] = torch.arange(128, device=dev, dtype=torch.uint8)
    rec = sd.fec_encode_frames(ctx, frames, R)
    allb = torch.cat([frames, rec], dim=1)
    keep = [i for i in range(160) if i not in set(range(1, 121, 5))][:128]
    rx = allb[:, keep].contiguous().reshape(Stx, F, 128, 512)
    idx = rx[:, :, :, 2].cpu().numpy()
    tx = sd.TxPipe(ctx, Stx, 4)
    tx.process(rx, idx)
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(20):  # clocks ramp
        y = tx.process(rx, idx)
    ctx.synchronize()
    ctx.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        y = tx.process(rx, idx)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    dms, dn = ctx.kernel_timing_read(K_FEC_DECODE)
    ims, inn = ctx.kernel_timing_read(K_INTERPOLATE)
    ctx.kernel_timing(False)
    print("   of which decode kernel %.3f ms, interpolator kernel %.3f ms per step" % (dms / max(dn, 1), ims / max(inn, 1)))
    nout = Stx * F * 16129 * 16
    print("tx pipe (decode 24 erasures + interpolate16) %8.3f ms / step  %8.1f Gsamples/s out  %7.1f GB/s (4.254 B/out)" %
          (dt * 1e3, nout / dt / 1e9, nout / dt / 1e9 * 4.254))
if "tx-random" in what:
    # config 4 under real loss: EVERY frame has its own random 24-erasure pattern (any of the 160 blocks), frames
    # resident on the device, block indices read from the headers by the planning kernel; no host work per frame
    import time

    import numpy as np

    F, R, Stx = 128, 32, 8
    frames = torch.randint(0, 256, (Stx * F, 128, 512), generator=g, device=dev, dtype=torch.uint8)
    frames[:, :, 2] = torch.arange(128, device=dev, dtype=torch.uint8)
    rec = sd.fec_encode_frames(ctx, frames, R)
    allb = torch.cat([frames, rec], dim=1)
    rs = np.random.RandomState(3)
    keep = np.stack([np.sort(rs.permutation(160)[:136])[:128] for _ in range(Stx * F)])  # 24 of 160 lost, first 128 arrivals
    rx = allb[torch.arange(Stx * F, device=dev)[:, None], torch.from_numpy(keep).to(dev)].contiguous().reshape(Stx, F, 128, 512)
    print("distinct loss patterns: %d of %d frames" % (len({k.tobytes() for k in keep}), Stx * F))
    tx = sd.TxPipe(ctx, Stx, 4)
    for _ in range(20):
        y = tx.process(rx)
    ctx.synchronize()
    ctx.kernel_timing(True)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        y = tx.process(rx)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    dms, dn = ctx.kernel_timing_read(K_FEC_DECODE)
    ims, inn = ctx.kernel_timing_read(K_INTERPOLATE)
    ctx.kernel_timing(False)
    nout = Stx * F * 16129 * 16
    print("   of which decode (plan + scatter + apply) %.3f ms, interpolator kernel %.3f ms per step" % (dms / max(dn, 1), ims / max(inn, 1)))
    print("tx pipe, a random 24-erasure pattern per frame (%d frames/step): %8.3f ms / step  %8.1f Gsamples/s out  %7.1f GB/s (4.254 B/out)" %
          (Stx * F, dt * 1e3, nout / dt / 1e9, nout / dt / 1e9 * 4.254))
    # the same frames, one shared pattern (round 1's best case)
    keep1 = np.tile(keep[:1], (Stx * F, 1))
    rx1 = allb[torch.arange(Stx * F, device=dev)[:, None], torch.from_numpy(keep1).to(dev)].contiguous().reshape(Stx, F, 128, 512)
    for _ in range(20):
        tx.process(rx1)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        tx.process(rx1)
    ctx.synchronize()
    dt1 = (time.perf_counter() - t0) / reps
    print("tx pipe, one pattern shared by all frames: %8.3f ms / step (ratio distinct / shared = %.2f)" % (dt1 * 1e3, dt / dt1))
if "host" in what:
    import time

    import numpy as np

    Sh, n = 8, 1 << 23
    xh = np.random.RandomState(1).randint(-32768, 32768, size=(Sh, n, 2)).astype(np.int16)
    rxp = sd.RxPipe(ctx, Sh, log2decim=4, nb_fec=32)
    rxp.process(xh)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        fr = rxp.process(xh)
    dt = (time.perf_counter() - t0) / reps
    print("rx pipe, HOST buffers (pageable numpy in, frames out): %8.2f ms / step  %7.2f Gsamples/s  (%5.1f GB/s over PCIe)" %
          (dt * 1e3, Sh * n / dt / 1e9, (xh.nbytes + fr.nbytes) / dt / 1e9))
    d1 = sd.Decimators(ctx, 1, 0)
    blk = xh[0, :65536]
    d1.decimate(4, 2, 16, blk)
    t0 = time.perf_counter()
    for _ in range(200):
        d1.decimate(4, 2, 16, blk)
    dt = (time.perf_counter() - t0) / 200
    print("single TestSource block (65536 samples, host pointers, one stream): %7.1f us / call  %6.3f Gsamples/s" % (dt * 1e6, 65536 / dt / 1e9))
