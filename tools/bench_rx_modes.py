#!/usr/bin/env python3
"""Rx step (8 x 2^25, decimate16_cen + framing + CM256 128+32) in the plumbing variants of sdrhip_rx_process:
immediate encode (three launches), pipelined with the encoder inside the decimator's launch, pipelined with separate launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrdaemon_amd as sd
import signals

ctx = sd.Context(0)
S, n = int(os.environ.get("STREAMS", "8")), 1 << int(os.environ.get("LOG2N", "25"))
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
ROUNDS = int(os.environ.get("ROUNDS", "3"))


def run(name, pipelined, fused, path="auto"):
    ctx.set_option("rx_fused", fused)
    ctx.set_option("decim_path", path)
    rx = sd.RxPipe(ctx, S, log2decim=4, nb_fec=32, pipelined=pipelined)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10):
            rx.process_view(x, 1, 0)
        torch.cuda.synchronize()
    K = 100
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        rx.process_view(x, i, 0)
    if pipelined:
        rx.flush_view()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    from sdrdaemon_amd.engine import K_DECIMATE, K_FEC_ENCODE
    ctx.kernel_timing(True)
    for i in range(20):
        rx.process_view(x, i, 0)
    d = ctx.kernel_timing_read(K_DECIMATE); e = ctx.kernel_timing_read(K_FEC_ENCODE)
    ctx.kernel_timing(False)
    print("   %s: decimate-class launch %.4f ms (n=%d), encode-class %.4f ms (n=%d)" % (name, d[0] / max(d[1], 1), d[1], e[0] / max(e[1], 1), e[1]))
    return ms


res = {}
ONLY_MODES = os.environ.get("MODES")  # e.g. MODES=immediate,overlap
for r in range(ROUNDS):
    for name, args in (("immediate (3 launches)", (False, 1)), ("pipelined, fused launch", (True, 1)), ("pipelined, separate launches", (True, 0)), ("pipelined, fused kernel without encoder units + encoder", (True, 2)),
                       ("pipelined, two streams (overlap)", (True, 3))):
        if ONLY_MODES and not any(k in name for k in ONLY_MODES.split(",")):
            continue
        res.setdefault(name, []).append(run(name, *args))
for k, v in res.items():
    print("%-32s %s ms/step  -> %.0f Gsamples/s" % (k, " ".join("%.4f" % t for t in v), S * n / min(v) / 1e6))
