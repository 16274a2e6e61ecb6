#!/usr/bin/env python3
"""K1m at decimate16: ONE wave per SIMD on an LDS-DMA ring of 4 groups (the product) against TWO waves per SIMD on rings of 2 groups
(mfma_ring = 2: 62 workgroups per XCD, spans half as long, the whole next group issued at the first step of the current one).
Rx step 8 x 2^25 (decimate16_cen + framing + CM256 128+32) and the decimator alone, interleaved rounds; outputs compared.
usage: python tools/experiments_r06/ring2_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrdaemon_amd as sd
import signals
from sdrdaemon_amd.engine import K_DECIMATE, K_FEC_ENCODE

ctx = sd.Context(0)
S, n = int(os.environ.get("STREAMS", "8")), 1 << int(os.environ.get("LOG2N", "25"))
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
ref = {}

def run(ring):
    ctx.set_option("mfma_ring", ring)
    rx = sd.RxPipe(ctx, S, log2decim=4, nb_fec=32, pipelined=False)
    fr = rx.process_view(x, 1, 0).torch().clone()
    torch.cuda.synchronize()
    if "frames" not in ref:
        ref["frames"] = fr
    same = bool(torch.equal(fr, ref["frames"]))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10):
            rx.process_view(x, 1, 0)
        torch.cuda.synchronize()
    K = 100
    t0 = time.perf_counter()
    for i in range(K):
        rx.process_view(x, i, 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    ctx.kernel_timing(True)
    for i in range(40):
        rx.process_view(x, i, 0)
    d = ctx.kernel_timing_read(K_DECIMATE); e = ctx.kernel_timing_read(K_FEC_ENCODE)
    ctx.kernel_timing(False)
    plan = rx.last_plan() if hasattr(rx, "last_plan") else None
    # the decimator alone (stream-order output)
    dd = sd.Decimators(ctx, S, 0)
    out = torch.empty((S, n >> 4, 2), dtype=torch.int16, device="cuda")
    for _ in range(20):
        dd.decimate(4, 2, 16, x, out=out)
    ctx.kernel_timing(True)
    for _ in range(40):
        dd.decimate(4, 2, 16, x, out=out)
    a = ctx.kernel_timing_read(K_DECIMATE)
    ctx.kernel_timing(False)
    if "dec" not in ref:
        ref["dec"] = out.clone()
    same2 = bool(torch.equal(out, ref["dec"]))
    return ms, d[0] / max(d[1], 1), e[0] / max(e[1], 1), a[0] / max(a[1], 1), same and same2, plan

for r in range(int(os.environ.get("ROUNDS", "4"))):
    for ring in (4, 2, 3):
        ms, k1, k3, alone, same, plan = run(ring)
        print("round %d  mfma_ring %d  Rx step %.4f ms  K1m %.4f  K3f %.4f   decimator alone %.4f ms   outputs equal to the first run's: %s  %s" % (r, ring, ms, k1, k3, alone, same, plan if r == 0 else ""))
        sys.stdout.flush()
