#!/usr/bin/env python3
"""HBM bandwidth probes on the GPU box (torch kernels): pure write (fill), pure read (sum), copy."""
import time
import torch

dev = torch.device("cuda", 0)
n = 1 << 29  # 2 GiB of int32
a = torch.empty(n, dtype=torch.int32, device=dev)
b = torch.empty(n, dtype=torch.int32, device=dev)


def timed(fn, reps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: a.fill_(7))
print("fill 2 GiB      %.3f ms  %.0f GB/s written" % (ms, 4 * n / ms / 1e6))
ms = timed(lambda: a.zero_())
print("memset 2 GiB    %.3f ms  %.0f GB/s written" % (ms, 4 * n / ms / 1e6))
ms = timed(lambda: b.copy_(a))
print("copy 2 GiB      %.3f ms  %.0f GB/s read + %.0f GB/s written" % (ms, 4 * n / ms / 1e6, 4 * n / ms / 1e6))
ms = timed(lambda: a.sum())
print("sum 2 GiB       %.3f ms  %.0f GB/s read" % (ms, 4 * n / ms / 1e6))
