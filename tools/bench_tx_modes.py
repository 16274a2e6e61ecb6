#!/usr/bin/env python3
"""Tx step (bench.py's configs[3]: 8 streams x 128 frames of config 3's output, a distinct 24-of-160 loss pattern per frame,
decode + interpolate16_cen) in the plumbing variants of sdrhip_tx_process: immediate (plan -> K4s -> K5w on one stream),
pipelined with the decode of batch N on the second stream beside the interpolator of batch N - 1, pipelined on one stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrdaemon_amd as sd
import signals
import headline_inputs as hi
from sdrdaemon_amd.engine import K_FEC_DECODE, K_INTERPOLATE

ctx = sd.Context(0)
S, n = 8, 1 << 25
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
meta = {"tv_sec": 1, "tv_usec": 0, "center_frequency_khz": 435000, "sample_rate": 625000, "nb_fec": 32}
rxf, keep = hi.tx_received_frames(ctx, x, meta)
del x
F = rxf.shape[1]
nout = S * F * 16129 << hi.TX_LOG2_INTERP
ROUNDS = int(os.environ.get("ROUNDS", "3"))
ctx.set_option("dec_max_rows", 32)
out = torch.empty((S, (F * 16129) << hi.TX_LOG2_INTERP, 2), dtype=torch.int16, device="cuda")


def run(name, pipelined, overlap):
    ctx.set_option("tx_overlap", overlap)
    tx = sd.TxPipe(ctx, S, hi.TX_LOG2_INTERP, pipelined=pipelined)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10):
            tx.process(rxf)
        torch.cuda.synchronize(); ctx.synchronize()
    K = 100
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        tx.process(rxf)
    if pipelined:
        tx.flush(device=rxf.device)
    torch.cuda.synchronize(); ctx.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    ctx.kernel_timing(True)
    for i in range(20):
        tx.process(rxf)
    d = ctx.kernel_timing_read(K_FEC_DECODE); e = ctx.kernel_timing_read(K_INTERPOLATE)
    ctx.kernel_timing(False)
    if pipelined:
        tx.flush(device=rxf.device)
    print("   %s: decode-class %.4f ms (n=%d), interpolate-class %.4f ms (n=%d)" % (name, d[0] / max(d[1], 1), d[1], e[0] / max(e[1], 1), e[1]), flush=True)
    return ms


res = {}
for r in range(ROUNDS):
    for name, args in (("immediate (plan, K4s, K5w on one stream)", (False, 1)), ("pipelined, two streams (overlap)", (True, 1)), ("pipelined, one stream", (True, 0))):
        res.setdefault(name, []).append(run(name, *args))
for k, v in res.items():
    print("%-44s %s ms/step  -> %.0f G output samples/s, pipe %.3f of 8 TB/s" % (k, " ".join("%.4f" % t for t in v), nout / min(v) / 1e6,
                                                                                   4.254 * nout / (min(v) * 1e-3) / 8e12))
