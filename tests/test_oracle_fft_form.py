"""The additive-FFT form of the CM256 128 + R encoder (sdrdaemon_amd/csrc/gf_encode128_fft.h) restated in numpy on the
ORACLE's field and checked against the oracle's cm256_encode (UDPSinkFEC.cpp:228-256): the facts the kernel is built on -- the subspace
polynomial of V7 is one constant on the coset 128 + V7, the fold onto 128 + V5, the split of the size-32 transform into two halves
of 16 behind its first stage, the zero constants of the first half's leading blocks.  CPU only; it pins the ALGORITHM to the oracle,
not the oracle to cm256cc (FEC parity stays unpinned, tests/test_oracle_vs_ref_cm256.py)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def field(oracle):
    mul = np.zeros((256, 256), np.uint8)
    for a in range(256):
        for b in range(a, 256):
            mul[a, b] = mul[b, a] = oracle.gf_mul(a, b)
    inv = np.zeros(256, np.uint8)
    for a in range(1, 256):
        inv[a] = oracle.gf_div(1, a)
    s = np.zeros((9, 256), np.uint8)  # s[k][x] = s_k(x): subspace polynomial of V_k = {0 .. 2^k - 1}
    s[0] = np.arange(256)
    for k in range(8):
        s[k + 1] = mul[s[k], s[k] ^ s[k][1 << k]]
    shat = np.stack([mul[s[k], inv[s[k][1 << k]]] for k in range(8)])  # normalised: shat_k(2^k) = 1
    return mul, inv, s, shat


def _ifft(mul, shat, vals, m, beta):
    d = vals.copy()
    for k in range(m):
        h = 1 << k
        for blk in range(0, 1 << m, 2 * h):
            c = int(shat[k][beta ^ blk])
            d[blk + h:blk + 2 * h] ^= d[blk:blk + h]
            d[blk:blk + h] ^= mul[c, d[blk + h:blk + 2 * h]]
    return d


def _fft(mul, shat, coef, m, beta, first_stage=None):
    d = coef.copy()
    for k in reversed(range(m) if first_stage is None else range(first_stage + 1)):
        h = 1 << k
        for blk in range(0, 1 << m, 2 * h):
            c = int(shat[k][beta ^ blk])
            d[blk:blk + h] ^= mul[c, d[blk + h:blk + 2 * h]]
            d[blk + h:blk + 2 * h] ^= d[blk:blk + h]
    return d


def test_subspace_polynomials(field):
    mul, inv, s, shat = field
    assert all(s[7][v] == 0 for v in range(128)) and len(set(s[7][128:].tolist())) == 1  # s_7 vanishes on V7, is ONE constant on 128 + V7
    for k in range(8):
        assert s[k][0] == 0 and shat[k][1 << k] == 1
        for a, b in ((3, 77), (128, 19), (200, 255)):  # linearised: additive
            assert s[k][a ^ b] == s[k][a] ^ s[k][b]
    # on the coset 128 + V5 the normalised s_5 and s_6 are constants: the fold of 128 coefficients onto 32
    assert len(set(shat[5][128:160].tolist())) == 1 and len(set(shat[6][128:160].tolist())) == 1
    # the leading block of every stage of the first half's transform has the constant zero (skipped by the kernel's first-half waves)
    assert all(shat[k][0] == 0 for k in range(8))


@pytest.mark.parametrize("R", [1, 2, 7, 13, 16, 31, 32])
def test_fft_form_equals_cm256_encode(oracle, field, R):
    mul, inv, s, shat = field
    rs = np.random.RandomState(100 + R)
    data = rs.randint(0, 256, (128, 508)).astype(np.uint8)
    exp = oracle.cm256_encode(data, R)
    q = int(s[7][128])
    c = 1
    for v in range(1, 128):
        c = int(mul[c, v])
    t5, t6 = int(shat[5][128]), int(shat[6][128])
    # the kernel's arrangement: two halves of 64 through the size-64 inverse transform (the size-128 stage has the constant 0 and
    # moves behind the fold), t5 fold per half, t6 across the halves, first stage of the size-32 transform, then two halves of 16
    lo = _ifft(mul, shat, data[:64], 6, 0)
    hi = _ifft(mul, shat, data[64:], 6, 64)
    elo = lo[:32] ^ mul[t5, lo[32:]]
    ehi = hi[:32] ^ mul[t5, hi[32:]]
    e = elo ^ mul[t6, ehi ^ elo]
    c4 = int(shat[4][128])
    a = e[:16] ^ mul[c4, e[16:]]
    b = e[16:] ^ a
    va = _fft(mul, shat, np.concatenate([a, b]), 5, 128, first_stage=3)  # stages 3..0: the halves do not meet any more
    par = np.bitwise_xor.reduce(data, axis=0)
    rec = np.stack([par ^ mul[int(mul[mul[r, c], inv[q]]), va[r]] for r in range(R)])
    assert np.array_equal(rec, exp)
    # and the textbook form: one size-128 inverse transform on V7, fold, one size-32 transform on 128 + V5
    coef = _ifft(mul, shat, data, 7, 0)
    e2 = coef[0:32] ^ mul[t5, coef[32:64]] ^ mul[t6, coef[64:96] ^ mul[t5, coef[96:128]]]
    assert np.array_equal(_fft(mul, shat, e2, 5, 128)[:R], va[:R])
