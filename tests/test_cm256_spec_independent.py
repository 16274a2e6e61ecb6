"""A THIRD derivation of CM256, sharing no line with the oracle (oracle/sdr_oracle.c) or the product (sdrdaemon_amd/csrc/gf256.cpp):
the round-5 review's point was that those two build their field from the same twenty recollected lines, so one wrong constant would
pass every test.  Here the code is written from the SPECIFICATION only -- two facts about upstream cm256 / gf256 (the library behind
cm256cc, called at UDPSinkFEC.cpp:246 and SDRdaemonFECBuffer.cpp:197):
  (A) the field polynomial is entry 3 of gf256's table of generator polynomials, which lists the sixteen degree-8 polynomials for which
      x is primitive in ascending order (stored as p >> 1) -- derived below by enumeration, never typed in;
  (B) recovery row i, column j of the encoder is (y_j + x_0) / (x_i + y_j) with x_i = OriginalCount + i, y_j = j (a Cauchy matrix
      normalised so that row 0 is all ones), recovery block i = sum_j a_ij * original_j.
Arithmetic is bit-serial carry-less multiplication and brute-force inversion (no logarithm tables, no generator), decoding is plain
Gaussian elimination on [I; A] (no closed-form Cauchy inverse, no LDU).  The oracle must agree with it byte for byte.
CPU only.  This makes the oracle's check independent of the oracle's own code; it does NOT pin either to cm256cc's sources, which are
not on this machine (FEC parity stays "unpinned": tests/test_oracle_vs_ref_cm256.py is the pin, skipped until they are mounted)."""
import numpy as np
import pytest


def _clmul_mod(a, b, poly):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        if a & 0x100:
            a ^= poly
        b >>= 1
    return r


def _x_is_primitive(poly):
    # x generates the multiplicative group iff its order is 255 (then the polynomial is irreducible as well)
    v, n = 2, 1
    while v != 1:
        v = _clmul_mod(v, 2, poly)
        n += 1
        if n > 255:
            return False
    return n == 255


@pytest.fixture(scope="module")
def spec():
    prim = [p for p in range(0x101, 0x200, 2) if _x_is_primitive(p)]
    assert len(prim) == 16  # phi(255) / 8
    poly = prim[3]          # fact (A)
    mul = np.zeros((256, 256), np.uint8)
    for a in range(256):
        for b in range(a, 256):
            mul[a, b] = mul[b, a] = _clmul_mod(a, b, poly)
    inv = np.zeros(256, np.uint8)
    for a in range(1, 256):
        inv[a] = int(np.nonzero(mul[a] == 1)[0][0])
    return poly, mul, inv


def _matrix(spec, k, m):
    _, mul, inv = spec
    a = np.zeros((m, k), np.uint8)
    for i in range(m):
        for j in range(k):
            a[i, j] = mul[j ^ k, inv[(k + i) ^ j]]  # fact (B)
    return a


def _encode(spec, x, m):
    _, mul, _ = spec
    a = _matrix(spec, x.shape[0], m)
    rec = np.zeros((m, x.shape[1]), np.uint8)
    for i in range(m):
        for j in range(x.shape[0]):
            rec[i] ^= mul[a[i, j]][x[j]]
    return rec


def _solve(spec, k, rows, data):
    """originals from any k of the k + m blocks: rows[t] = block index (< k: original, >= k: recovery row - k), data[t] its bytes."""
    _, mul, inv = spec
    m = max([r - k + 1 for r in rows if r >= k] + [1])
    a = _matrix(spec, k, m)
    g = np.zeros((k, k), np.uint8)
    for t, r in enumerate(rows):
        if r < k:
            g[t, r] = 1
        else:
            g[t] = a[r - k]
    g, d = g.copy(), data.copy()
    for c in range(k):  # Gauss-Jordan over GF(2^8)
        p = next(t for t in range(c, k) if g[t, c])
        if p != c:
            g[[c, p]] = g[[p, c]]
            d[[c, p]] = d[[p, c]]
        s = inv[g[c, c]]
        g[c] = mul[s][g[c]]
        d[c] = mul[s][d[c]]
        for t in range(k):
            if t != c and g[t, c]:
                f = g[t, c]
                g[t] ^= mul[f][g[c]]
                d[t] ^= mul[f][d[c]]
    return d


def test_polynomial_is_the_fourth_primitive_one(spec, oracle):
    poly, mul, inv = spec
    assert poly == 0x14D
    # ... and the oracle's field is this field, element by element
    for a in (0, 1, 2, 3, 0x53, 0x80, 0xCA, 0xFF):
        for b in range(256):
            assert oracle.gf_mul(a, b) == mul[a, b], (a, b)
            if b:
                assert oracle.gf_div(a, b) == mul[a, inv[b]], (a, b)


def test_matrix_is_the_normalised_cauchy_matrix(spec, oracle):
    for k, m in ((128, 32), (128, 128), (5, 3), (2, 2), (200, 56)):
        a = _matrix(spec, k, m)
        assert np.all(a[0] == 1)  # row 0 = XOR parity (what DecodeM1 and UDPSinkFEC's single-block case rely on)
        for i in (0, 1, m // 2, m - 1):
            for j in (0, 1, k // 2, k - 1):
                assert oracle.matrix_element(k + i, k, j) == a[i, j], (k, m, i, j)


@pytest.mark.parametrize("k,m,bb", [(128, 32, 24), (128, 1, 8), (128, 128, 8), (5, 3, 33), (2, 2, 16), (16, 16, 64)])
def test_oracle_encode_equals_the_specification(spec, oracle, k, m, bb):
    rs = np.random.RandomState(1000 + k + m)
    x = rs.randint(0, 256, size=(k, bb)).astype(np.uint8)
    assert np.array_equal(oracle.cm256_encode(x, m), _encode(spec, x, m))


@pytest.mark.parametrize("case", ["config4_stride5", "worst32", "random24_any_rows", "block0_and_last"])
def test_cross_decoding(spec, oracle, case):
    """blocks encoded by one implementation, decoded by the other: oracle -> Gaussian elimination, specification -> oracle's cm256_decode."""
    rs = np.random.RandomState(77)
    k, m, bb = 128, 32, 16
    x = rs.randint(0, 256, size=(k, bb)).astype(np.uint8)
    if case == "config4_stride5":
        erased, use = list(range(1, 121, 5)), list(range(24))
    elif case == "worst32":
        erased, use = list(range(0, 128, 4)), list(range(31, -1, -1))
    elif case == "random24_any_rows":
        erased = sorted(rs.choice(k, 24, replace=False).tolist())
        use = sorted(rs.choice(m, 24, replace=False).tolist())
    else:
        erased, use = [0, 127], [7, 30]
    keep = [j for j in range(k) if j not in erased]
    rows = keep + [k + r for r in use]
    # oracle encodes, the specification solves
    rec_o = oracle.cm256_encode(x, m)
    got = _solve(spec, k, rows, np.concatenate([x[keep], rec_o[use]]))
    assert np.array_equal(got, x), case
    # the specification encodes, the oracle decodes (in place, cm256_decode's contract: recovery rows become the erased originals)
    rec_s = _encode(spec, x, m)
    data = np.concatenate([x[keep], rec_s[use]]).copy()
    rc, idx2 = oracle.cm256_decode(data, np.array(rows), k, m)
    assert rc == 0
    out = np.zeros_like(x)
    for row, i in zip(data, idx2):
        out[i] = row
    assert np.array_equal(out, x), case


# ------------------------------------------------------------------ the PRODUCT's host-side planning against the same specification
# (gf256.cpp is plain host C++ inside libsdrhip.so: tables and matrices the kernels consume.  No GPU call here; the per-byte block
# arithmetic of the kernels is checked against the oracle in the -m gpu tests, the oracle against the specification above.)
@pytest.fixture(scope="module")
def product_host():
    import ctypes as C
    import __graft_entry__ as g

    g.build()
    from sdrdaemon_amd import _lib

    lib = _lib.lib()
    f = {}
    f["tables"] = getattr(lib, "_ZN6sdrhip15gf_build_tablesEPh")
    f["tables"].argtypes, f["tables"].restype = [C.c_void_p], C.c_int
    f["matrix"] = getattr(lib, "_ZN6sdrhip19cm256_encode_matrixEiiPh")
    f["matrix"].argtypes, f["matrix"].restype = [C.c_int, C.c_int, C.c_void_p], None
    f["plan"] = getattr(lib, "_ZN6sdrhip17cm256_decode_planEiiPKhPiPhS3_S3_")
    f["plan"].argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]
    f["plan"].restype = C.c_int
    return f


def test_product_multiplier_tables_are_this_field(spec, product_host):
    _, mul, _ = spec
    tab = np.zeros((256, 32), np.uint8)
    product_host["tables"](tab.ctypes.data)
    for m in range(256):
        assert np.array_equal(tab[m, 0:8], mul[m, np.arange(8)]), m             # m * (low three bits)
        assert np.array_equal(tab[m, 8:16], mul[m, np.arange(8) << 3]), m       # m * (middle three bits)
        assert np.array_equal(tab[m, 16:20], mul[m, np.arange(4) << 6]), m      # m * (top two bits)


@pytest.mark.parametrize("k,m", [(128, 32), (128, 128), (5, 3), (2, 2), (200, 56), (255, 1)])
def test_product_encode_matrix_is_the_specification_s(spec, product_host, k, m):
    got = np.zeros((m, k), np.uint8)
    product_host["matrix"](k, m, got.ctypes.data)
    assert np.array_equal(got, _matrix(spec, k, m))


@pytest.mark.parametrize("case", ["config4_stride5", "worst32_any_order", "random24_any_rows"])
def test_product_decode_plan_solves_the_specification_s_system(spec, product_host, case):
    """cm256_decode_plan's coefficient rows applied to blocks the SPECIFICATION encoded give back the erased originals."""
    import ctypes as C

    _, mul, _ = spec
    rs = np.random.RandomState(5)
    k, m, bb = 128, 32, 12
    x = rs.randint(0, 256, size=(k, bb)).astype(np.uint8)
    rec = _encode(spec, x, m)
    if case == "config4_stride5":
        erased, use = list(range(1, 121, 5)), list(range(24))
    elif case == "worst32_any_order":
        erased, use = list(range(3, 128, 4)), rs.permutation(32).tolist()
    else:
        erased = sorted(rs.choice(k, 24, replace=False).tolist())
        use = sorted(rs.choice(m, 24, replace=False).tolist())
    keep = [j for j in range(k) if j not in erased]
    order = rs.permutation(k)  # arrival order: originals and recovery blocks mixed (SDRdaemonFECBuffer.cpp:143-170)
    idx = np.array(keep + [k + r for r in use], np.uint8)[order]
    data = np.concatenate([x[keep], rec[use]])[order]
    n_rec = C.c_int(0)
    rec_pos, er, coef = np.zeros(k, np.uint8), np.zeros(256, np.uint8), np.zeros((k, k), np.uint8)
    assert product_host["plan"](k, m, idx.ctypes.data, C.byref(n_rec), rec_pos.ctypes.data, er.ctypes.data, coef.ctypes.data) == 0
    n = n_rec.value
    assert n == len(erased) and sorted(er[:n].tolist()) == erased
    c = coef.reshape(-1)[: n * k].reshape(n, k)
    for i in range(n):
        v = np.zeros(bb, np.uint8)
        for p in range(k):
            if c[i, p]:
                v ^= mul[c[i, p]][data[p]]
        assert np.array_equal(v, x[er[i]]), (case, i)
        assert idx[rec_pos[i]] >= k  # restored INTO a recovery descriptor (cm256_decode's in-place contract)
