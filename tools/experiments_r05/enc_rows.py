#!/usr/bin/env python3
"""CM256 encode of 2048 frames (128 originals) for small numbers of recovery blocks: the generic matrix kernel against the additive-FFT
encoder (ctx option enc_min_rows) -- where is the crossover?  -> profiles/r05_enc_rows.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrdaemon_amd as sd
ctx = sd.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
F = 2048
frames = torch.randint(0, 256, (F, 128, 512), generator=g, device="cuda", dtype=torch.uint8)
def t(R):
    for _ in range(5): sd.fec_encode_frames(ctx, frames, R)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): sd.fec_encode_frames(ctx, frames, R)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 50 * 1e3
for R in (1, 2, 3, 4, 5, 6, 8, 10, 12, 13, 16, 24, 32):
    ctx.set_option("enc_min_rows", 32 if R < 32 else 1); a = t(R); ra = sd.fec_encode_frames(ctx, frames[:64], R).clone()
    ctx.set_option("enc_min_rows", 1); b = t(R); rb = sd.fec_encode_frames(ctx, frames[:64], R).clone()
    print("R = %2d: generic matrix kernel %.4f ms, additive FFT %.4f ms per %d frames (same bytes: %s)" % (R, a, b, F, bool(torch.equal(ra, rb))))
