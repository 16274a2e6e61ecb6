#!/usr/bin/env python3
"""Feasibility probe: does the 128-original encoder run beside the matrix-core decimator when the two are enqueued on
different HIP streams (two contexts)?  Prints sequential and concurrent wall times per pair of launches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
from sdrdaemon_amd.engine import fec_encode_frames  # noqa: E402

dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ca, cb = sd.Context(0, stream=s1), sd.Context(0, stream=s2)
g = torch.Generator(device=dev).manual_seed(1)
S, n, L = 8, 1 << 25, 4
x = torch.randint(-32768, 32768, (S, n, 2), generator=g, device=dev, dtype=torch.int16)
out = torch.empty((S, n >> L, 2), dtype=torch.int16, device=dev)
frames = torch.randint(0, 256, (1040, 128, 512), generator=g, device=dev, dtype=torch.uint8)
ca.set_option("decim_path", "mfma")
d = sd.Decimators(ca, S, 0)
K = 60


def run(span, mode):
    ca.set_option("mfma_span", span)
    for warm in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            if mode in ("dec", "both"):
                d.decimate(L, 2, 16, x, out=out)
            if mode in ("enc", "both"):
                fec_encode_frames(cb, frames, 32)
        ca.synchronize(); cb.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
    return dt


for span in (0, 32768):
    a, b, c = run(span, "dec"), run(span, "enc"), run(span, "both")
    print("span %6d: decimate alone %.4f ms, encode alone %.4f ms, both enqueued on two streams %.4f ms per pair (sum %.4f)" % (span, a, b, c, a + b), flush=True)
