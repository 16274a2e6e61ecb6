"""Oracle vs the committed golden vectors (reference outputs): runs everywhere, incl. the GPU box
where neither /root/reference nor a compiler for it exists."""
import hashlib

import numpy as np
import pytest

import signals
from golden_util import Golden

G = Golden()


@pytest.mark.parametrize("flavour", ["eo1", "db"])
def test_oracle_matches_golden(oracle, flavour):
    n = 0
    for case in G.cases:
        if case["flavour"] != flavour:
            continue
        x = G.input(case)
        pos, outs = 0, []
        if case["kind"] == "decimate":
            d = oracle.decimators(case["bias"])
            ss_out = None
            for c in case["chunks"]:
                o, ss_out = d.decimate(case["log2"], case["fcpos"], case["sample_size"], x[pos:pos + c])
                outs.append(o)
                pos += c
            assert ss_out == case["sample_size_out"], case["key"]
        else:
            u = oracle.interpolators()
            for c in case["chunks"]:
                outs.append(u.interpolate(case["log2"], x[pos:pos + c]))
                pos += c
        assert np.array_equal(np.concatenate(outs), G.expected(case)), case["key"]
        n += 1
    assert n == 168


def test_oracle_matches_big_digests(oracle):
    for b in G.big:
        x = signals.noise(1 << 20, b["seed"])
        if b["kind"] == "decimate16_cen_blocks":
            d = oracle.decimators(b["bias"])
            h = hashlib.sha256()
            for i in range(16):
                o, _ = d.decimate(4, 2, 16, x[i * 65536:(i + 1) * 65536])
                h.update(o.tobytes())
            assert h.hexdigest() == b["sha256"]
        else:
            y = oracle.interpolators().interpolate(4, x[:b["n"]])
            assert hashlib.sha256(y.tobytes()).hexdigest() == b["sha256"]


def test_oracle_matches_long_golden(oracle):
    """centred cascades on 65536 samples (the cases the matrix-core kernel is checked against on the GPU)"""
    GL = Golden("dsp_golden_long")
    for case in GL.cases:
        x = GL.input(case)
        d = oracle.decimators(case["bias"])
        pos, outs = 0, []
        for c in case["chunks"]:
            o, ss = d.decimate(case["log2"], 2, 16, x[pos:pos + c])
            outs.append(o)
            pos += c
        assert np.array_equal(np.concatenate(outs), GL.expected(case)), case["key"]
    assert len(GL.cases) == 60


def test_hash_noise_numpy_equals_torch():
    """the counter-based input generator of the headline goldens: numpy (golden generation) == torch (GPU tests, bench.py)"""
    import torch

    for seed, n, start in ((1000, 100003, 0), (4000, 4097, 1 << 26), (1063, 1 << 16, 12345)):
        a = signals.hash_noise(n, seed, start)
        b = signals.hash_noise_torch(n, seed, "cpu", start).numpy()
        assert np.array_equal(a, b)
        assert np.array_equal(a, signals.hash_noise(n + start, seed)[start:]) if start < (1 << 20) else True
    a = signals.hash_noise(1 << 18, 1000)
    assert a.min() < -32000 and a.max() > 32000 and abs(float(a.mean())) < 200.0  # full scale, centred


def test_framer_stamps_follow_the_sample_clock(oracle):
    """the product's time-stamp rule restated in the oracle framer: a frame opened p samples into a write() call is stamped
    floor(p * 10^6 / sample_rate) us after the call's stamp; CRC over the stamped record (zlib cross-check)"""
    import zlib

    x = signals.noise(3 * 16129 + 500, 21)
    fr = oracle.framer(nb_fec_blocks=8, tv_sec=10, tv_usec=999990, sample_rate=625000)
    frames = fr.write(x)
    exp = [(10, 999990), (11, 25796), (11, 51602)]  # 16129 / 625000 s = 25806.4 us
    for f in range(3):
        meta = frames[f, 0, 4:28].tobytes()
        assert (int.from_bytes(meta[12:16], "little"), int.from_bytes(meta[16:20], "little")) == exp[f]
        assert int.from_bytes(meta[20:24], "little") == (zlib.crc32(meta[:20]) & 0xFFFFFFFF)
    # second call: the open frame keeps its stamp, the next frame counts from THIS call's first sample
    fr.s.tv_sec, fr.s.tv_usec = 20, 5
    more = fr.write(signals.noise(2 * 16129, 22))
    meta = more[1, 0, 4:28].tobytes()
    p = 16129 - 500
    assert (int.from_bytes(meta[12:16], "little"), int.from_bytes(meta[16:20], "little")) == (20, 5 + p * 1000000 // 625000)
    # literal reference behaviour: every frame a call opens carries the stamp as given
    lit = oracle.framer(nb_fec_blocks=8, tv_sec=10, tv_usec=999990, stamp_from_samples=0).write(x)
    assert all(lit[f, 0, 16:24].tobytes() == lit[0, 0, 16:24].tobytes() for f in range(3))


def test_oracle_chain_matches_headline_digests(oracle):
    """the headline goldens (reference decimator + framer / encoder restatement) against the oracle's own decimator on two
    streams of config 5's bank: the whole-output digests the GPU tests and bench.py compare with"""
    from golden_util import headline

    H = headline()
    m, b = H["meta"], H["bank64"]
    for k in (0, 63):
        x = signals.hash_noise(1 << b["log2n"], b["seeds"][k])
        y, ss = oracle.decimators(0).decimate(4, 2, 16, x)
        assert hashlib.sha256(y.tobytes()).hexdigest() == b["dec_sha256"][k]
        fr = oracle.framer(nb_fec_blocks=m["nb_fec"], tv_sec=m["tv_sec"], tv_usec=m["tv_usec"], sample_rate=m["sample_rate"],
                           center_frequency_khz=m["center_frequency_khz"])
        frames = fr.write(y)
        h = hashlib.sha256()
        for f in range(frames.shape[0]):
            h.update(frames[f].tobytes())
            h.update(oracle.frame_encode(frames[f], m["nb_fec"]).tobytes())
        assert frames.shape[0] == b["nframes"][k] and h.hexdigest() == b["frames_sha256"][k]
