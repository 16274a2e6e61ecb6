#!/usr/bin/env python3
"""Prototype (CPU, numpy): CM256's 128-original Cauchy encode as an additive FFT (Lin-Chung-Han novel basis) instead of a
32 x 128 matrix product.  recovery_r = Par ^ r * S(128 ^ r),  S(x) = sum_j d_j / (x ^ j)  (gf_encode128_body.h),  and
S(x) = P(x) / Q(x) with Q = the subspace polynomial of V7 = {0..127} (constant q = Q(128) on the coset 128 + V7) and P the
polynomial of degree < 128 with P(j) = c d_j, c = Q'(0) = product of the nonzero elements of V7.  So: IFFT_128 at 0 of the data,
fold the 128 novel-basis coefficients onto the coset 128 + V5, FFT_32, scale row r by r c / q.  Checked against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
orc = Oracle()
mul = np.zeros((256, 256), np.uint8)
for a in range(256):
    for b in range(256):
        mul[a, b] = orc.gf_mul(a, b)
inv = np.zeros(256, np.uint8)
for a in range(1, 256):
    inv[a] = orc.gf_div(1, a)

def s_tab():
    """s[k][x] = s_k(x) (subspace polynomial of V_k = {0..2^k-1}) for all x; shat[k][x] = s_k(x) / s_k(2^k)"""
    s = np.zeros((9, 256), np.uint8)
    s[0] = np.arange(256)
    for k in range(8):
        sk_vk = s[k][1 << k]
        # s_{k+1}(x) = s_k(x) * s_k(x ^ v_k) = s_k(x) * (s_k(x) ^ s_k(v_k))
        s[k + 1] = mul[s[k], s[k] ^ sk_vk]
    shat = np.zeros((8, 256), np.uint8)
    for k in range(8):
        shat[k] = mul[s[k], inv[s[k][1 << k]]]
    return s, shat
S, SH = s_tab()
assert all(S[7][v] == 0 for v in range(128)) and len(set(S[7][128:].tolist())) == 1
q = int(S[7][128])
c = 1
for v in range(1, 128):
    c = int(mul[c, v])

def ifft(vals, m, beta):
    """values on beta + V_m (natural order) -> novel-basis coefficients; vals: (2^m, n) uint8"""
    d = vals.copy()
    for k in range(m):  # stage k: blocks of 2^(k+1)
        h = 1 << k
        for blk in range(0, 1 << m, 2 * h):
            cst = int(SH[k][beta ^ blk])
            for i in range(blk, blk + h):
                d[i + h] ^= d[i]
                if cst:
                    d[i] ^= mul[cst, d[i + h]]
    return d

def fft(coef, m, beta):
    d = coef.copy()
    for k in reversed(range(m)):
        h = 1 << k
        for blk in range(0, 1 << m, 2 * h):
            cst = int(SH[k][beta ^ blk])
            for i in range(blk, blk + h):
                if cst:
                    d[i] ^= mul[cst, d[i + h]]
                d[i + h] ^= d[i]
    return d

rs = np.random.RandomState(1)
data = rs.randint(0, 256, (128, 508)).astype(np.uint8)
R = 32
exp = orc.cm256_encode(data, R)
# sanity of the transform pair
assert np.array_equal(fft(ifft(data, 7, 0), 7, 0), data)
coef = ifft(data, 7, 0)                      # P_unscaled(j) = d_j
t5, t6 = int(SH[5][128]), int(SH[6][128])    # constant on 128 + V5
e = coef[0:32] ^ mul[t5, coef[32:64]] ^ mul[t6, coef[64:96] ^ mul[t5, coef[96:128]]]
ev = fft(e, 5, 128)                          # P_unscaled(128 ^ r), r = 0..31
par = np.bitwise_xor.reduce(data, axis=0)
rec = np.zeros((R, 508), np.uint8)
for r in range(R):
    k = int(mul[mul[r, c], inv[q]])
    rec[r] = par ^ mul[k, ev[r]]
print("LCH encode == oracle cm256_encode:", np.array_equal(rec, exp))
nm = sum(1 for k in range(7) for blk in range(0, 128, 2 << k) if SH[k][blk]) 
print("q", q, "c", c, "t5", t5, "t6", t6)

# ---- the kernel's arrangement (gf_encode128_fft.h): two halves of 64 through one IFFT-64 routine, the size-128 stage (constant 0)
# folded into the t6 step by linearity, then FFT-32 on the coset 128 + V5
def ifft64(vals, beta):
    return ifft(vals, 6, beta)
e_lo = None
for h in (0, 1):
    d = ifft64(data[64 * h:64 * h + 64], 64 * h)
    t = d[0:32] ^ mul[t5, d[32:64]]
    if h == 0:
        e_lo = t
    else:
        e2 = e_lo ^ mul[t6, t ^ e_lo]
ev2 = fft(e2, 5, 128)
rec2 = np.stack([par ^ mul[int(mul[mul[r, c], inv[q]]), ev2[r]] for r in range(R)])
print("kernel arrangement == oracle:", np.array_equal(rec2, exp))
