#!/usr/bin/env python3
"""The drop-in mode as sdrdaemonrx / sdrdaemontx see it: one TestSource-sized block (65 536 samples) per synchronous call through
host pointers (SDRHIP_MEM_HOST): microseconds per call, against the same call on device-resident data; the asynchronous entries
(sdrhip_rx_submit / collect, sdrhip_tx_submit / collect).  usage: python tools/bench_host_block.py [rx|tx|all]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402

ctx = sd.Context(0)
WHAT = sys.argv[1] if len(sys.argv) > 1 else "all"


def tx_section():
    """Tx side: received frames from host memory (a distinct 24-of-160 loss pattern per frame), decode + interpolate16_cen:
    the synchronous per-frame call (what UDPSourceFEC::read + Upsampler::process amount to), synchronous batches of 8, and the
    asynchronous entry with batches of 8 / 32 frames, 3 batches in flight"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import signals
    import headline_inputs as hi

    x = torch.stack([signals.hash_noise_torch(1 << 25, 1000, "cuda")])
    meta = {"tv_sec": 1, "tv_usec": 0, "center_frequency_khz": 435000, "sample_rate": 625000, "nb_fec": 32}
    rxf, _ = hi.tx_received_frames(ctx, x, meta)
    del x
    rxh = rxf[0].cpu().numpy()  # (128 frames, 128, 512)
    F = rxh.shape[0]
    ctx.set_option("dec_max_rows", 32)
    for L in (4, 0):
        tx = sd.TxPipe(ctx, 1, L)
        for f in range(8):
            tx.process(rxh[f:f + 1])
        t0 = time.perf_counter()
        for f in range(F):
            tx.process(rxh[f:f + 1])
        t1 = (time.perf_counter() - t0) / F * 1e6
        t0 = time.perf_counter()
        for f in range(0, F, 8):
            tx.process(rxh[f:f + 8])
        t8 = (time.perf_counter() - t0) / F * 1e6
        print("tx pipe (interp x%d), host pointers, synchronous: %7.1f us per frame one frame per call, %7.1f us per frame in calls of 8" % (1 << L, t1, t8), flush=True)
        for pinned in (False, True):
            src = rxh
            if pinned:
                src = ctx.host_alloc(rxh.shape, np.uint8)
                src[:] = rxh
            for bf in (8, 32):
                p = sd.TxPipe(ctx, 1, L)
                p.set_async(depth=4)

                def run(rounds):
                    inflight = 0
                    for r in range(rounds):
                        for f in range(0, F, bf):
                            p.submit(src[f:f + bf])
                            inflight += 1
                            if inflight == 3:
                                p.collect(wait=True)
                                inflight -= 1
                    while inflight:
                        p.collect(wait=True)
                        inflight -= 1
                run(1)
                R = 4
                t0 = time.perf_counter()
                run(R)
                ta = (time.perf_counter() - t0) / (R * F) * 1e6
                print("tx submit / collect (interp x%d), %2d frames per batch, %s: %7.1f us per frame = %5.1f x the synchronous per-frame call, %7.1f M output samples/s" %
                      (1 << L, bf, "in place (pinned source)" if pinned else "staged (pageable source) ", ta, t1 / ta, (16129 << L) / ta), flush=True)
            if pinned:
                ctx.host_free(src)
    ctx.set_option("dec_max_rows", 128)


if WHAT in ("tx", "all"):
    tx_section()
if WHAT == "tx":
    sys.exit(0)
for n in (65536, 262144, 1 << 20):
    x = np.random.default_rng(1).integers(-32768, 32768, (n, 2), dtype=np.int16)
    xd = torch.from_numpy(x).cuda()
    for L in (4,):
        d = sd.Decimators(ctx, 1, 0)
        for _ in range(50):
            d.decimate(L, 2, 16, x)
        t0 = time.perf_counter()
        K = 400
        for _ in range(K):
            d.decimate(L, 2, 16, x)
        th = (time.perf_counter() - t0) / K * 1e6
        out = torch.empty((n >> L, 2), dtype=torch.int16, device="cuda")
        for _ in range(50):
            d.decimate(L, 2, 16, xd, out=out)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            d.decimate(L, 2, 16, xd, out=out)
            ctx.synchronize()
        td = (time.perf_counter() - t0) / K * 1e6
        print("decimate%d_cen, %7d samples per call: host pointers %7.1f us (%6.1f M samples/s)   device pointers + sync %7.1f us" % (1 << L, n, th, n / th, td), flush=True)

# the Rx pipe (decimate16_cen + framing + CM256 128+32) fed from host memory, one stream
for n in (65536, 262144, 1 << 20, 1 << 22, 1 << 24):
    x = np.random.default_rng(2).integers(-32768, 32768, (n, 2), dtype=np.int16)
    for pipelined in (False, True):
        rx = sd.RxPipe(ctx, 1, log2decim=4, nb_fec=32, pipelined=pipelined)
        K = max(8, min(400, (1 << 26) // n))
        for _ in range(min(K, 20)):
            rx.process(x, 1, 2)
        t0 = time.perf_counter()
        for _ in range(K):
            rx.process(x, 1, 2)
        th = (time.perf_counter() - t0) / K * 1e6
        print("rx pipe, %8d samples per call, host pointers%s: %8.1f us per call = %7.1f M samples/s" % (n, " (pipelined)" if pipelined else "", th, n / th), flush=True)

# the asynchronous entry (sdrhip_rx_submit / sdrhip_rx_collect): 65 536-sample blocks, k per upload + launch + download, the
# collector one batch behind the submitter; staged (pageable source -> memcpy into pinned) and in place (sdrhip_host_alloc)
n = 65536
for pinned in (False, True):
    for blocks in (1, 4, 8, 16, 32):
        nb = 16 * blocks
        if pinned:
            src = ctx.host_alloc((1, nb * n, 2))
            src[:] = np.random.default_rng(3).integers(-32768, 32768, (1, nb * n, 2), dtype=np.int16)
        else:
            src = np.random.default_rng(3).integers(-32768, 32768, (1, nb * n, 2), dtype=np.int16)
        rx = sd.RxPipe(ctx, 1, log2decim=4, nb_fec=32)
        rx.set_async(depth=4, blocks=blocks)

        blks = [src[:, b * n:(b + 1) * n] for b in range(nb)]

        def run(rounds):
            inflight = 0
            for r in range(rounds):
                for b in range(nb):
                    rx.submit(blks[b], 1, 2)
                    if (b + 1) % blocks == 0:
                        inflight += 1
                        if inflight == 3:
                            rx.collect(wait=True, max_frames=blocks * 65536 // (16129 * 16) + 2)
                            inflight -= 1
            while inflight:
                rx.collect(wait=True, max_frames=blocks * 65536 // (16129 * 16) + 2)
                inflight -= 1

        run(1)
        R = max(1, 64 // blocks)
        t0 = time.perf_counter()
        run(R)
        dt = time.perf_counter() - t0
        print("rx submit / collect, 65536-sample blocks, %2d per batch, %s: %7.1f us per block = %7.1f M samples/s" %
              (blocks, "in place (pinned source)" if pinned else "staged (pageable source) ", dt / (R * nb) * 1e6, R * nb * n / dt / 1e6), flush=True)
        if pinned:
            ctx.host_free(src)
