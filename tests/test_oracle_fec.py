"""FEC half of the oracle.  PARITY UNPINNED (cm256cc absent): pinned by algebraic known-answer
tests instead: field identities for polynomial 0x14D, recovery row 0 = XOR parity,
Cauchy element formula, MDS round trips, invertibility of sampled sub-matrices, CRC-32 = zlib."""
import zlib

import numpy as np
import pytest


def test_gf256_field_identities(oracle):
    assert oracle.lib.orc_gf_exp(0) == 1 and oracle.lib.orc_gf_exp(1) == 2
    assert oracle.gf_mul(0x80, 2) == 0x4D  # xtime(0x80) = 0x100 ^ 0x14D
    exps = [oracle.lib.orc_gf_exp(i) for i in range(255)]
    assert sorted(exps) == list(range(1, 256))  # generator 2 is primitive for 0x14D
    for a in range(1, 256):
        assert oracle.lib.orc_gf_exp(oracle.lib.orc_gf_log(a)) == a
        inv = oracle.gf_div(1, a)
        assert oracle.gf_mul(a, inv) == 1
        assert oracle.gf_mul(a, 1) == a and oracle.gf_mul(a, 0) == 0
    rs = np.random.RandomState(1)
    for a, b, c in rs.randint(0, 256, size=(500, 3)):
        a, b, c = int(a), int(b), int(c)
        assert oracle.gf_mul(a, b) == oracle.gf_mul(b, a)
        assert oracle.gf_mul(a, oracle.gf_mul(b, c)) == oracle.gf_mul(oracle.gf_mul(a, b), c)
        assert oracle.gf_mul(a, b ^ c) == oracle.gf_mul(a, b) ^ oracle.gf_mul(a, c)
        if b:
            assert oracle.gf_mul(oracle.gf_div(a, b), b) == a


def test_matrix_element_formula(oracle):
    for r in range(0, 128, 7):
        for j in range(0, 128, 5):
            m = oracle.matrix_element(128 + r, 128, j)
            assert m == oracle.gf_div(j ^ 128, (128 + r) ^ j)
            if r == 0:
                assert m == 1


def test_crc32_is_zlib(oracle):
    rs = np.random.RandomState(3)
    for n in (0, 1, 20, 24, 511):
        b = rs.randint(0, 256, n).astype(np.uint8).tobytes()
        assert oracle.crc32(b) == (zlib.crc32(b) & 0xFFFFFFFF)


def test_encode_row0_is_xor_parity(oracle):
    rs = np.random.RandomState(4)
    x = rs.randint(0, 256, size=(128, 508)).astype(np.uint8)
    rec = oracle.cm256_encode(x, 32)
    assert np.array_equal(rec[0], np.bitwise_xor.reduce(x, axis=0))
    # linearity of every row
    y = rs.randint(0, 256, size=(128, 508)).astype(np.uint8)
    assert np.array_equal(oracle.cm256_encode(x ^ y, 32), rec ^ oracle.cm256_encode(y, 32))


def _roundtrip(oracle, x, rec, erased, use_rows):
    """Erase originals `erased`, deliver survivors in index order followed by recovery rows use_rows."""
    k = x.shape[0]
    keep = [i for i in range(k) if i not in set(erased)]
    data = np.concatenate([x[keep], rec[use_rows]]).copy()
    idx = np.array(keep + [k + r for r in use_rows])
    rc, idx2 = oracle.cm256_decode(data, idx, k, len(use_rows))
    assert rc == 0
    out = np.zeros_like(x)
    for row, i in zip(data, idx2):
        assert i < k
        out[i] = row
    return out


@pytest.mark.parametrize("case", ["stride5_24", "first32", "last32", "random24", "one_row0", "two"])
def test_mds_roundtrip(oracle, case):
    rs = np.random.RandomState(11)
    x = rs.randint(0, 256, size=(128, 508)).astype(np.uint8)
    rec = oracle.cm256_encode(x, 32)
    if case == "stride5_24":
        erased, rows = list(range(1, 121, 5)), list(range(24))
    elif case == "first32":
        erased, rows = list(range(32)), list(range(32))
    elif case == "last32":
        erased, rows = list(range(96, 128)), list(range(31, -1, -1))
    elif case == "random24":
        erased = sorted(rs.choice(128, 24, replace=False).tolist())
        rows = sorted(rs.choice(32, 24, replace=False).tolist())
    elif case == "one_row0":
        erased, rows = [77], [0]
    else:
        erased, rows = [0, 127], [5, 17]
    assert np.array_equal(_roundtrip(oracle, x, rec, erased, rows), x)


def test_decode_m1_quirk_is_mirrored(oracle):
    """RecoveryCount == 1 takes upstream's XOR shortcut whatever the row."""
    rs = np.random.RandomState(12)
    x = rs.randint(0, 256, size=(128, 508)).astype(np.uint8)
    rec = oracle.cm256_encode(x, 4)
    out = _roundtrip(oracle, x, rec, [9], [2])
    assert not np.array_equal(out[9], x[9])  # wrong by design, like the reference + library
    expected = rec[2] ^ np.bitwise_xor.reduce(np.delete(x, 9, axis=0), axis=0)
    assert np.array_equal(out[9], expected)


def test_small_geometry(oracle):
    rs = np.random.RandomState(13)
    for k, m, bb in ((2, 2, 16), (5, 3, 33), (16, 16, 64)):
        x = rs.randint(0, 256, size=(k, bb)).astype(np.uint8)
        rec = oracle.cm256_encode(x, m)
        erased = sorted(rs.choice(k, min(m, k), replace=False).tolist())
        rows = list(range(len(erased)))
        if len(rows) == 1:
            rows = [0]
        assert np.array_equal(_roundtrip(oracle, x, rec, erased, rows), x)


def test_framer_layout_and_fecbuffer_roundtrip(oracle):
    import signals

    x = signals.noise(3 * 16129 + 500, 21)
    fr = oracle.framer(nb_fec_blocks=8, tv_sec=1, tv_usec=2)
    frames = np.concatenate([fr.write(x[:10000]), fr.write(x[10000:40000]), fr.write(x[40000:])])
    assert frames.shape == (3, 128, 512)
    for f in range(3):
        assert np.array_equal(frames[f, :, 0], np.full(128, f, np.uint8))  # frameIndex lo
        assert np.array_equal(frames[f, :, 2], np.arange(128, dtype=np.uint8))  # blockIndex
        meta = frames[f, 0, 4:28].tobytes()
        assert int.from_bytes(meta[0:4], "little") == 435000 and meta[10] == 128 and meta[11] == 8
        assert int.from_bytes(meta[20:24], "little") == (zlib.crc32(meta[:20]) & 0xFFFFFFFF)
        assert not frames[f, 0, 28:].any()
        payload = frames[f, 1:, 4:].reshape(-1).view(np.int16).reshape(-1, 2)
        assert np.array_equal(payload, x[f * 16129:(f + 1) * 16129])
    # loss + recovery through the SDRdaemonFECBuffer restatement
    buf = oracle.fecbuffer()
    outs = []
    for f in range(3):
        rec = oracle.frame_encode(frames[f], 8)
        blocks = list(frames[f]) + list(rec)
        lost = {3, 50, 127, 0, 64, 99}  # 6 of the originals, 8 recovery available
        for i, sb in enumerate(blocks):
            if i in lost:
                continue
            o = buf.write_and_read(sb)
            if o is not None:
                outs.append(o)
    o = buf.write_and_read(np.zeros(512, np.uint8) + 255)  # next frame index flushes frame 2
    outs.append(o)
    assert len(outs) == 4 and not outs[0].any()  # the very first emission is the empty slot
    for f in range(3):
        got = outs[f + 1].view(np.int16).reshape(-1, 2)
        assert np.array_equal(got, x[f * 16129:(f + 1) * 16129])
