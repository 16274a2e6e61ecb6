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
    main()
