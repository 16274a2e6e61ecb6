#!/usr/bin/env python3
"""FFT encoder (K3f) stand-alone against the number of frames per launch: does the launch follow the fullest CU (1040 frames = 4.06 per
CU: 16 CUs carry five)?  Second column: the same launch behind a kernel that has just written the frames (the Rx step's situation).
usage: python tools/experiments_r06/enc_count_scan.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sdrdaemon_amd as sd
from sdrdaemon_amd.engine import K_FEC_ENCODE
import time
ctx = sd.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
FMAX = 4096
allf = torch.randint(0, 256, (FMAX, 128, 512), generator=g, device=dev, dtype=torch.uint8)
allf[:, :, 2] = torch.arange(128, device=dev, dtype=torch.uint8)
src = allf.clone()
def timed(fn, reps=40):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        fn(); ctx.synchronize()
    ctx.kernel_timing(True)
    for _ in range(reps):
        fn()
    ms, n = ctx.kernel_timing_read(K_FEC_ENCODE)
    ctx.kernel_timing(False)
    return ms / max(n, 1)
UNITS = os.environ.get("ENC_UNITS", "frame")  # frame | half (context option enc_units)
ctx.set_option("enc_units", UNITS)
print("enc_units =", UNITS)
for rnd in range(2):
    for F in (256, 512, 768, 1024, 1040, 1152, 1280, 1296, 1536, 2048, 2560, 4096):
        fr = allf[:F]
        a = timed(lambda: sd.fec_encode_frames(ctx, fr, 32))
        def fresh():
            fr.copy_(src[:F])
            sd.fec_encode_frames(ctx, fr, 32)
        b = timed(fresh)
        print("round %d  frames %5d (%.2f per CU)  encode %.4f ms = %.2f us per frame-per-CU   behind a fresh write of the frames %.4f ms" % (rnd, F, F / 256, a, a * 1e3 / (F / 256), b))
