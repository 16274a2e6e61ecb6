"""Deterministic int16 IQ test inputs (SURVEY.md 8d / BASELINE.md 3.4).

All generators return (n, 2) int16 arrays (interleaved I, Q like IQSample,
SDRDaemon.h:52-70).  np.random.RandomState is numpy's frozen legacy MT19937
stream, so seeds reproduce across numpy versions; golden fixtures nevertheless
store their inputs.
"""
import numpy as np


def cw(n, amplitude=3276.8, df=100e3, fs=10e6, start=0):
    """TestSource-like CW (TestSource.cpp:395-422 shape) in double precision:
    I = round(A cos phi_n), Q = round(A sin phi_n), phi_n = 2 pi n df / fs."""
    k = np.arange(start, start + n, dtype=np.float64)
    ph = 2.0 * np.pi * k * (df / fs)
    iq = np.empty((n, 2), dtype=np.int16)
    iq[:, 0] = np.clip(np.rint(amplitude * np.cos(ph)), -32768, 32767).astype(np.int16)
    iq[:, 1] = np.clip(np.rint(amplitude * np.sin(ph)), -32768, 32767).astype(np.int16)
    return iq


def noise(n, seed=1234, bits=16):
    """Uniform full-scale random samples of `bits` effective bits (8 / 12 / 16)."""
    rs = np.random.RandomState(seed)
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1))
    return rs.randint(lo, hi, size=(n, 2)).astype(np.int16)


def const(n, value=32767):
    return np.full((n, 2), value, dtype=np.int16)


def alternating(n):
    """+32767 / -32768 alternation: forces int32 wrap-around in stages 3-4."""
    a = np.empty((n, 2), dtype=np.int16)
    a[0::2] = 32767
    a[1::2] = -32768
    return a


def impulse(n, pos=0, value=32767):
    a = np.zeros((n, 2), dtype=np.int16)
    a[pos] = (value, -value)
    return a


def zeros(n):
    return np.zeros((n, 2), dtype=np.int16)


def mixed(n, seed=7):
    """CW + noise bursts + full-scale steps: exercises carries and both signs."""
    rs = np.random.RandomState(seed)
    a = cw(n, amplitude=20000.0, df=137e3, fs=2.4e6).astype(np.int32)
    a += rs.randint(-9000, 9000, size=(n, 2))
    k = max(n // 7, 1)
    a[k:k + max(n // 50, 1)] = 32767
    a[3 * k:3 * k + max(n // 50, 1)] = -32768
    return np.clip(a, -32768, 32767).astype(np.int16)


ALL = {
    "cw_small": lambda n: cw(n, 3276.8),
    "cw_full": lambda n: cw(n, 32767.0),
    "noise": lambda n: noise(n, 1234),
    "const": lambda n: const(n),
    "alternating": alternating,
    "impulse": lambda n: impulse(n, 5),
    "zeros": zeros,
    "mixed": mixed,
}


# ---- counter-based noise with a torch twin: the same full-scale uniform int16 IQ stream on the CPU (numpy: golden
# generation from the compiled reference) and on the GPU (torch: the `-m gpu` tests and bench.py make GiB-sized inputs
# in milliseconds, no upload).  Sample i of stream `seed` = lowbias32(i ^ mix(seed)), I = low half, Q = high half.
_HM = 0xFFFFFFFF


def _seed_mix(seed):
    return (int(seed) * 0x9E3779B1 + 0x85EBCA6B) & _HM


def hash_noise(n, seed, start=0):
    """(n, 2) int16, numpy.  Chunked so that the int64 temporaries stay small."""
    out = np.empty((n, 2), dtype=np.int16)
    flat = out.view(np.uint32).reshape(n)
    sm = np.int64(_seed_mix(seed))
    step = 1 << 22
    for a in range(0, n, step):
        b = min(n, a + step)
        x = (np.arange(start + a, start + b, dtype=np.int64) ^ sm) & _HM
        x ^= x >> 16
        x = (x * 0x7FEB352D) & _HM
        x ^= x >> 15
        x = (x * 0x846CA68B) & _HM
        x ^= x >> 16
        flat[a:b] = x.astype(np.uint32)
    return out


def hash_noise_torch(n, seed, device, start=0):
    """(n, 2) int16 torch tensor on `device`, bit-identical to hash_noise(n, seed, start)."""
    import torch

    out = torch.empty((n, 2), dtype=torch.int16, device=device)
    flat = out.view(torch.int32).reshape(n)
    sm = _seed_mix(seed)
    step = 1 << 24
    for a in range(0, n, step):
        b = min(n, a + step)
        x = (torch.arange(start + a, start + b, dtype=torch.int64, device=device) ^ sm) & _HM
        x ^= x >> 16
        x = (x * 0x7FEB352D) & _HM
        x ^= x >> 15
        x = (x * 0x846CA68B) & _HM
        x ^= x >> 16
        flat[a:b] = x.to(torch.int32)  # (wraps modulo 2^32: the same 32 bits)
    return out
