#!/usr/bin/env python3
"""Generates tests/golden/dsp_golden.npz from the REAL reference (oracle/_ref, i.e.
/root/reference/sdmnbase/{Decimators,Interpolators,HBFilterTraits}.cpp compiled by
oracle/Makefile with the reference's own flags).  Run in the build container only:

    make -C oracle && python tests/golden/make_golden.py

The fixture holds data only: seeded int16 IQ inputs and the reference's outputs (both the
USE_SSE4_1 = EO1 and the DB flavour), plus SHA-256 digests of large runs.  Call chunking is
part of each case (state carries across the calls of a case)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import signals  # noqa: E402
from oracle_lib import Reference  # noqa: E402

CHUNKS = [4096, 1024 + 16, 3072 - 16, 8192]  # sums to 16384


def main():
    arrays, index = {}, []
    inputs = {
        "mixed": signals.mixed(16384, 7),
        "alternating": signals.alternating(16384),
        "noise": signals.noise(16384, 1234),
        "cw_full": signals.cw(16384, 32767.0),
        "zeros": signals.zeros(16384),
    }
    for k, v in inputs.items():
        arrays["in_" + k] = v
    for flav in ("eo1", "db"):
        ref = Reference(flav)
        for sig in inputs:
            for ss0 in ((16,) if sig != "noise" else (8, 12, 16)):
                x = inputs[sig] if ss0 == 16 else (inputs[sig] >> (16 - ss0)).astype(np.int16)
                for fcpos in (0, 1, 2):
                    for log2 in range(0, 7):
                        d = ref.decimators()
                        outs, pos, ss_out = [], 0, None
                        for c in CHUNKS:
                            o, ss_out = d.decimate(log2, fcpos, ss0, x[pos:pos + c])
                            outs.append(o)
                            pos += c
                        key = "dec_%s_%s_ss%d_fc%d_L%d" % (flav, sig, ss0, fcpos, log2)
                        arrays[key] = np.concatenate(outs)
                        index.append({"key": key, "kind": "decimate", "flavour": flav, "bias": ref.bias,
                                      "input": sig, "sample_size": ss0, "sample_size_out": ss_out,
                                      "fcpos": fcpos, "log2": log2, "chunks": CHUNKS})
        for sig in ("mixed", "noise", "alternating"):
            x = inputs[sig][:1024]
            for log2 in range(0, 7):
                u = ref.interpolators()
                outs, pos = [], 0
                for c in (256, 1, 255, 512):
                    outs.append(u.interpolate(log2, x[pos:pos + c]))
                    pos += c
                key = "int_%s_%s_L%d" % (flav, sig, log2)
                arrays[key] = np.concatenate(outs)
                index.append({"key": key, "kind": "interpolate", "flavour": flav, "input": sig,
                              "log2": log2, "chunks": [256, 1, 255, 512], "n_in": 1024})
    # large runs: digest only (inputs regenerated from tests/signals.py by the test)
    big = []
    for flav in ("eo1", "db"):
        ref = Reference(flav)
        x = signals.noise(1 << 20, 4321)
        d = ref.decimators()
        h = hashlib.sha256()
        for b in range(16):  # 16 TestSource-sized blocks of 65536 (TestSource.h:33)
            o, _ = d.decimate(4, 2, 16, x[b * 65536:(b + 1) * 65536])
            h.update(o.tobytes())
        big.append({"kind": "decimate16_cen_blocks", "flavour": flav, "bias": ref.bias, "seed": 4321,
                    "n": 1 << 20, "block": 65536, "sha256": h.hexdigest()})
        u = ref.interpolators()
        y = u.interpolate(4, x[:16129 * 4])
        big.append({"kind": "interpolate16_cen", "flavour": flav, "seed": 4321, "n": 16129 * 4,
                    "sha256": hashlib.sha256(y.tobytes()).hexdigest()})
    np.savez_compressed(os.path.join(HERE, "dsp_golden.npz"), **arrays)
    with open(os.path.join(HERE, "dsp_golden.json"), "w") as f:
        json.dump({"cases": index, "big": big}, f, indent=0)
    print("cases:", len(index), "arrays:", len(arrays))
    fec_golden()
    long_golden()
    headline_golden()
    tx_headline_golden()


def long_golden():
    """Centred cascades on inputs long enough for the matrix-core kernel to engage at EVERY ratio (decimate64 needs 4096 + 8
    spans of 4096 samples): 65536 samples, one call and a ragged two-call split, both flavours -> dsp_golden_long.npz."""
    arrays, index = {}, []
    inputs = {"noise": signals.noise(65536, 77), "mixed": signals.mixed(65536, 78), "alternating": signals.alternating(65536)}
    for k, v in inputs.items():
        arrays["in_" + k] = v
    for flav in ("eo1", "db"):
        ref = Reference(flav)
        for sig, x in inputs.items():
            for log2 in range(2, 7):
                for chunks in ([65536], [40000 + 16, 25536 - 16]):
                    d = ref.decimators()
                    outs, pos = [], 0
                    for c in chunks:
                        o, ss_out = d.decimate(log2, 2, 16, x[pos:pos + c])
                        outs.append(o)
                        pos += c
                    key = "declong_%s_%s_L%d_c%d" % (flav, sig, log2, len(chunks))
                    arrays[key] = np.concatenate(outs)
                    index.append({"key": key, "kind": "decimate", "flavour": flav, "bias": ref.bias, "input": sig, "sample_size": 16,
                                  "sample_size_out": ss_out, "fcpos": 2, "log2": log2, "chunks": chunks})
    np.savez_compressed(os.path.join(HERE, "dsp_golden_long.npz"), **arrays)
    with open(os.path.join(HERE, "dsp_golden_long.json"), "w") as f:
        json.dump({"cases": index}, f, indent=0)
    print("long cases:", len(index))


HEADLINE_META = {"tv_sec": 1, "tv_usec": 0, "center_frequency_khz": 435000, "sample_rate": 625000, "nb_fec": 32}


def _headline_stream(ref, orc, n, seed, x=None):
    """one stream of the benchmark: signals.hash_noise(n, seed) (or the samples handed in) -> the REFERENCE's decimate16_cen (EO1
    build, one call) -> UDPSinkFEC framing + CM256 128+32 (the oracle's framer with the product's time-stamp rule, and its encoder).
    -> (sha256 of the decimated stream, sha256 of the finished frames [f][160][512], number of frames)"""
    if x is None:
        x = signals.hash_noise(n, seed)
    y, ss = ref.decimators().decimate(4, 2, 16, x)
    assert ss == 16 and y.shape[0] == n >> 4
    m = HEADLINE_META
    fr = orc.framer(nb_fec_blocks=m["nb_fec"], tv_sec=m["tv_sec"], tv_usec=m["tv_usec"], center_frequency_khz=m["center_frequency_khz"],
                    sample_rate=m["sample_rate"], sample_bytes=2, sample_bits=16)
    frames = fr.write(y)
    h = hashlib.sha256()
    for f in range(frames.shape[0]):
        h.update(frames[f].tobytes())
        h.update(orc.frame_encode(frames[f], m["nb_fec"]).tobytes())
    return hashlib.sha256(y.tobytes()).hexdigest(), h.hexdigest(), int(frames.shape[0])


def headline_golden():
    """Whole-output digests of the BENCHMARKED launches: bench.py's 8 streams x 2^25 samples, configs[2] as one
    stream of 2^27, and config 5's bank of 64 streams (x 2^22 here) -- inputs from signals.hash_noise (torch twin on the GPU),
    decimated by the compiled reference itself."""
    from oracle_lib import Oracle

    ref, orc = Reference("eo1"), Oracle()
    out = {"meta": HEADLINE_META, "flavour": "eo1"}
    for name, seeds, log2n in (("bank8", list(range(1000, 1008)), 25), ("one27", [4000], 27), ("bank64", list(range(1000, 1064)), 22)):
        dec, frm, nfr = [], [], []
        for seed in seeds:
            a, b, c = _headline_stream(ref, orc, 1 << log2n, seed)
            dec.append(a); frm.append(b); nfr.append(c)
            print("headline", name, seed, a[:12], b[:12], c, flush=True)
        out[name] = {"seeds": seeds, "log2n": log2n, "dec_sha256": dec, "frames_sha256": frm, "nframes": nfr}
    with open(os.path.join(HERE, "headline_golden.json"), "w") as f:
        json.dump(out, f, indent=0)


def _headline_stream_job(args):
    from oracle_lib import Oracle

    n, seed = args
    return seed, _headline_stream(Reference("eo1"), Oracle(), n, seed)


def headline64_golden(procs=8):
    """config 5's bank at the benchmark's step size (round 5: one layout across N): 64 streams x 2^25 samples, the job
    `bench.py --gpus N` runs for every N > 1 and, as an extra `configs` line, at N = 1.  Same recipe as headline_golden (compiled
    reference decimate16_cen -> oracle framer + encoder), one process per stream.  -> key "bank64_25" of headline_golden.json
    (the first 8 streams are bank8's)."""
    import multiprocessing as mp

    path = os.path.join(HERE, "headline_golden.json")
    with open(path) as f:
        out = json.load(f)
    seeds, log2n = list(range(1000, 1064)), 25
    res = {}
    for i, seed in enumerate(out["bank8"]["seeds"]):  # (already made, by the same function)
        res[seed] = (out["bank8"]["dec_sha256"][i], out["bank8"]["frames_sha256"][i], out["bank8"]["nframes"][i])
    todo = [(1 << log2n, seed) for seed in seeds if seed not in res]
    with mp.Pool(procs) as pool:
        for seed, r in pool.imap_unordered(_headline_stream_job, todo):
            res[seed] = r
            print("headline bank64_25", seed, r[0][:12], r[1][:12], r[2], flush=True)
    out["bank64_25"] = {"seeds": seeds, "log2n": log2n, "dec_sha256": [res[s][0] for s in seeds],
                        "frames_sha256": [res[s][1] for s in seeds], "nframes": [res[s][2] for s in seeds]}
    with open(path, "w") as f:
        json.dump(out, f, indent=0)


def _ts_stream_job(args):
    import headline_inputs as hi
    from oracle_lib import Oracle

    n, seed = args
    orc = Oracle()
    x = hi.ts_oracle_samples(orc, n, seed)
    return seed, hashlib.sha256(x.tobytes()).hexdigest(), _headline_stream(Reference("eo1"), orc, n, seed, x=x)


def ts_headline_golden(procs=8):
    """The headline step on the input every BASELINE config names (VERDICT r5 #3a): bench.py --input testsource -- 8 streams x 2^25
    samples of the TestSource bank's integer NCO (tests/headline_inputs.py: 10 Msps, -20 dB CW at +100 kHz + 1 kHz x stream), made
    HERE by the oracle's restatement of that NCO, through the compiled reference's decimate16_cen and the oracle framer + encoder.
    -> key "ts_bank8" of headline_golden.json (input digests included: the bank's samples themselves are checked too)."""
    import multiprocessing as mp

    path = os.path.join(HERE, "headline_golden.json")
    with open(path) as f:
        out = json.load(f)
    seeds, log2n = list(range(1000, 1008)), 25
    res = {}
    with mp.Pool(procs) as pool:
        for seed, xs, r in pool.imap_unordered(_ts_stream_job, [(1 << log2n, seed) for seed in seeds]):
            res[seed] = (xs,) + r
            print("ts headline", seed, xs[:12], r[0][:12], r[1][:12], r[2], flush=True)
    import headline_inputs as hi
    out["ts_bank8"] = {"seeds": seeds, "log2n": log2n, "config": [hi.ts_config_string(s) for s in seeds],
                       "input_sha256": [res[s][0] for s in seeds], "dec_sha256": [res[s][1] for s in seeds],
                       "frames_sha256": [res[s][2] for s in seeds], "nframes": [res[s][3] for s in seeds]}
    with open(path, "w") as f:
        json.dump(out, f, indent=0)


def tx_headline_golden():
    """Whole-output digests of the BENCHMARKED Tx launch: bench.py's configs[3] -- the first 128 frames of
    every stream of the bank8 Rx run above (reference decimator -> oracle framer + encoder), 24 of 160 blocks lost per frame
    (tests/headline_inputs.py: tx_keep_sets), the oracle's cm256_decode, then the REFERENCE's interpolate16_cen over the
    stream of recovered frames.  -> key "tx_bank8" of headline_golden.json."""
    import headline_inputs as hi
    from oracle_lib import Oracle

    ref, orc = Reference("eo1"), Oracle()
    path = os.path.join(HERE, "headline_golden.json")
    with open(path) as f:
        out = json.load(f)
    m = out["meta"]
    seeds, log2n, F = out["bank8"]["seeds"], out["bank8"]["log2n"], hi.TX_FRAMES
    keep = hi.tx_keep_sets(len(seeds) * F)
    pay_sha, iq_sha, nrec_hist = [], [], {}
    for s, seed in enumerate(seeds):
        x = signals.hash_noise(1 << log2n, seed)
        y, _ = ref.decimators().decimate(4, 2, 16, x)
        fr = orc.framer(nb_fec_blocks=m["nb_fec"], tv_sec=m["tv_sec"], tv_usec=m["tv_usec"], center_frequency_khz=m["center_frequency_khz"],
                        sample_rate=m["sample_rate"], sample_bytes=2, sample_bits=16)
        frames = fr.write(y)[:F]
        assert frames.shape[0] == F
        payload = np.zeros((F, 127, 508), np.uint8)
        for f in range(F):
            allb = np.concatenate([frames[f], orc.frame_encode(frames[f], m["nb_fec"])])  # (160, 512)
            k = keep[s * F + f]
            data = np.ascontiguousarray(allb[k][:, 4:])
            n_rec = int((k >= 128).sum())
            nrec_hist[n_rec] = nrec_hist.get(n_rec, 0) + 1
            rc, idx = orc.cm256_decode(data, k.astype(np.uint8), 128, n_rec)
            assert rc == 0
            got = np.zeros((128, 508), np.uint8)
            seen = np.zeros(128, bool)
            for i in range(128):
                got[idx[i]] = data[i]
                seen[idx[i]] = True
            assert seen.all() and np.array_equal(got, frames[f][:, 4:]), "the oracle's decode must restore the frame"
            payload[f] = got[1:]
        iq = payload.reshape(-1).view(np.int16).reshape(-1, 2)
        assert np.array_equal(iq, y[:F * 16129])
        z = ref.interpolators().interpolate(hi.TX_LOG2_INTERP, iq)
        pay_sha.append(hashlib.sha256(payload.tobytes()).hexdigest())
        iq_sha.append(hashlib.sha256(z.tobytes()).hexdigest())
        print("tx headline", seed, pay_sha[-1][:12], iq_sha[-1][:12], z.shape, flush=True)
    out["tx_bank8"] = {"seeds": seeds, "log2n": log2n, "frames": F, "keep_seed": hi.TX_KEEP_SEED, "log2interp": hi.TX_LOG2_INTERP,
                       "payload_sha256": pay_sha, "iq_sha256": iq_sha, "recovery_blocks_used_histogram": {str(k): v for k, v in sorted(nrec_hist.items())}}
    with open(path, "w") as f:
        json.dump(out, f, indent=0)


def fec_golden():
    """FEC vectors from the REAL cm256cc library (oracle/_ref/libsdrref_cm256.so, built by `make -C oracle
    LIBCM256CCSRC=<dir>` where the library's sources exist).  Absent on this image: nothing is written and the FEC
    half of the oracle stays unpinned (tests/test_oracle_vs_ref_cm256.py skips, tests/test_oracle_golden.py has no
    FEC cases).  With it: recovery blocks for R = 1, 8, 32, 128 and decoded frames for the SURVEY 8d loss sets."""
    import ctypes as C

    from oracle_lib import ORACLE_DIR

    lib = os.path.join(ORACLE_DIR, "_ref", "libsdrref_cm256.so")
    if not os.path.exists(lib):
        print("fec golden: cm256cc not available (oracle/_ref/libsdrref_cm256.so missing) - FEC parity stays unpinned")
        return
    L = C.CDLL(lib)
    L.sdrref_cm256_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.sdrref_cm256_decode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    orig = np.ascontiguousarray(signals.noise(128 * 127, 900).view(np.uint8).reshape(128, 508))
    arrays, index = {"fec_orig": orig}, []
    for R in (1, 8, 32, 128):
        rec = np.zeros((R, 508), np.uint8)
        assert L.sdrref_cm256_encode(128, R, 508, orig.ctypes.data, rec.ctypes.data) == 0
        arrays["fec_rec_R%d" % R] = rec
        index.append({"kind": "cm256_encode", "k": 128, "R": R, "key": "fec_rec_R%d" % R})
    allb = np.concatenate([orig, arrays["fec_rec_R32"]])
    rs = np.random.RandomState(11)
    for name, lost in (("fixed24", set(range(1, 121, 5))), ("random24", set(rs.choice(160, 24, replace=False).tolist()) | {0}),
                       ("m1_row128", {7}), ("m1_row129", {7, 128})):
        got = [i for i in range(160) if i not in lost][:128]
        n_rec = sum(1 for i in got if i >= 128)
        data, idx = np.ascontiguousarray(allb[got]), np.array(got, np.uint8)
        rc = L.sdrref_cm256_decode(128, n_rec, 508, data.ctypes.data, idx.ctypes.data)
        arrays["fec_dec_%s_data" % name], arrays["fec_dec_%s_idx" % name] = data, idx
        index.append({"kind": "cm256_decode", "name": name, "received": got, "rc": int(rc), "key": "fec_dec_%s" % name})
    np.savez_compressed(os.path.join(HERE, "fec_golden.npz"), **arrays)
    with open(os.path.join(HERE, "fec_golden.json"), "w") as f:
        json.dump({"cases": index}, f, indent=0)
    print("fec golden cases:", len(index))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "headline":
        headline_golden()
        tx_headline_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "headline64":
        headline64_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "ts_headline":
        ts_headline_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "tx_headline":
        tx_headline_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "fec":
        fec_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "long":
        long_golden()
    else:
        main()
