#!/usr/bin/env python3
"""(the "with the dispatch" variants need tools/experiments_r06/ktimer_ext_launch.patch applied to the library; without it they are skipped)
What the kernel-class timers cost the headline step (round 6): the same 8 x 2^25 step with no timers, with marker-packet event
pairs around every launch (ktime_ext = 0), with the events handed to the dispatch (hipExtLaunchKernelGGL, ktime_ext = 1), at
strides 1 and 4; interleaved rounds; also: do the two methods agree on the kernel time?
usage: python tools/experiments_r06/timer_cost.py [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
import signals  # noqa: E402
from sdrdaemon_amd.engine import K_DECIMATE, K_FEC_ENCODE  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = sd.Context(0)
S, n = 8, 1 << 25
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
rx = sd.RxPipe(ctx, S, log2decim=4, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=32, center_frequency_khz=435000, sample_rate=625000)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    for _ in range(10):
        rx.process_view(x, 1, 0)
    torch.cuda.synchronize()


HAVE_EXT = True
try:
    ctx.set_option("ktime_ext", 1)
except Exception:  # (the product library: ktimer_ext_launch.patch is not applied)
    HAVE_EXT = False


def run(timing, ext, stride, steps=100):
    if HAVE_EXT:
        ctx.set_option("ktime_ext", ext)
    if isinstance(stride, tuple):  # (decimator stride, everything else)
        ctx.set_option("ktime_stride", stride[1])
        ctx.set_option("ktime_stride_class", "%d:%d" % (K_DECIMATE, stride[0]))
    else:
        ctx.set_option("ktime_stride", stride)
    for _ in range(10):
        rx.process_view(x, 1, 0)
    torch.cuda.synchronize()
    ctx.kernel_timing(timing)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rx.process_view(x, 1, 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    per = {}
    if timing:
        for c in (K_DECIMATE, K_FEC_ENCODE):
            t, k = ctx.kernel_timing_read(c)
            per[c] = (t / max(k, 1), k)
    ctx.kernel_timing(False)
    return ms, per


variants = [("no timers", False, 1, 1), ("markers, every launch", True, 0, 1), ("with the dispatch, every launch", True, 1, 1),
            ("markers, every 4th", True, 0, 4), ("with the dispatch, every 4th", True, 1, 4), ("markers, K1m every launch, K3f every 4th", True, 0, (1, 4))]
if not HAVE_EXT:
    variants = [v for v in variants if v[2] == 0 or not v[1]]
for r in range(rounds):
    for name, timing, ext, stride in variants:
        ms, per = run(timing, ext, stride)
        extra = ""
        if per:
            extra = "   K1m %.4f ms (%d)   K3f %.4f ms (%d)" % (per[K_DECIMATE][0], per[K_DECIMATE][1], per[K_FEC_ENCODE][0], per[K_FEC_ENCODE][1])
        print("round %d  %-34s step %.4f ms%s" % (r, name, ms, extra), flush=True)
