#!/usr/bin/env python3
"""The box's two states (K1m 0.218 / 0.226 ms per 2^28 samples in consecutive processes, profiles/r06_k1m_sched_strategy.txt): is it WHERE the
buffers lie?  One process, NI input banks and NO output buffers allocated side by side, decimate16_cen (8 x 2^25) timed on every (input, output)
pair, three rounds.  If the time follows the pair, physical placement is the state; if it follows the round, it is the box.
usage: python tools/experiments_r06/placement_probe.py [NI] [NO]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrdaemon_amd as sd
import signals
from sdrdaemon_amd.engine import K_DECIMATE

NI = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NO = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = sd.Context(0)
S, n = 8, 1 << 25
ins, outs, pads = [], [], []
for i in range(NI):
    ins.append(torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)]))
    pads.append(torch.empty(((i + 1) * 3 << 20) + 4096 * (i + 1), dtype=torch.uint8, device="cuda"))  # shifts what follows
for j in range(NO):
    outs.append(torch.empty((S, n >> 4, 2), dtype=torch.int16, device="cuda"))
    pads.append(torch.empty(((j + 1) * 5 << 20) + 8192 * (j + 1), dtype=torch.uint8, device="cuda"))
print("inputs at", [hex(t.data_ptr()) for t in ins])
print("outputs at", [hex(t.data_ptr()) for t in outs])
d = sd.Decimators(ctx, S, 0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    d.decimate(4, 2, 16, ins[0], out=outs[0]); ctx.synchronize()
for rnd in range(3):
    for i in range(NI):
        row = []
        for j in range(NO):
            for _ in range(5):
                d.decimate(4, 2, 16, ins[i], out=outs[j])
            ctx.kernel_timing(True)
            for _ in range(30):
                d.decimate(4, 2, 16, ins[i], out=outs[j])
            ms, k = ctx.kernel_timing_read(K_DECIMATE)
            ctx.kernel_timing(False)
            row.append(ms / max(k, 1))
        print("round %d  input %d  x outputs 0..%d:  %s" % (rnd, i, NO - 1, "  ".join("%.4f" % v for v in row)), flush=True)
