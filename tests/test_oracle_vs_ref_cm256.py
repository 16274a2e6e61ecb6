"""Pins the FEC half of the oracle to the REAL cm256cc library -- the day it exists on the machine.

cm256cc (f4exb/cm256cc, the fork of catid/cm256 the reference links: cm256cc/CMakeLists.txt:12-34) is not part of the
reference tree and is absent from this image, so this module SKIPS and the oracle's CM256 stays "parity unpinned".
With the sources mounted, `make -C oracle LIBCM256CCSRC=<dir>` compiles them unchanged into
oracle/_ref/libsdrref_cm256.so (recipe: oracle/Makefile, shim: oracle/ref_cm256_shim.cpp) and these tests diff the
restatement (oracle/sdr_oracle.c) against it; tests/golden/make_golden.py then freezes FEC vectors."""
import ctypes as C
import os

import numpy as np
import pytest

import signals
from oracle_lib import ORACLE_DIR

LIB = os.path.join(ORACLE_DIR, "_ref", "libsdrref_cm256.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="cm256cc is not on this machine: FEC parity unpinned (see module docstring)")


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(LIB)
    L.sdrref_cm256_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.sdrref_cm256_decode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert L.sdrref_cm256_initialized() == 1
    return L


def ref_encode(L, orig, m):
    k, bb = orig.shape
    rec = np.zeros((m, bb), np.uint8)
    o = np.ascontiguousarray(orig)
    assert L.sdrref_cm256_encode(k, m, bb, o.ctypes.data, rec.ctypes.data) == 0
    return rec


def test_encode_every_recovery_count(oracle, ref):
    """the reference's geometry (128 originals of 508 bytes, UDPSinkFEC.cpp:195-246), every R in 1..128"""
    for R in range(1, 129):
        orig = signals.noise(128 * 127, 700 + R).view(np.uint8).reshape(128, 508)
        assert np.array_equal(oracle.cm256_encode(orig, R), ref_encode(ref, orig, R)), R


@pytest.mark.parametrize("k,m,bb", [(1, 1, 16), (2, 3, 1), (5, 5, 33), (100, 156, 64), (200, 56, 508), (255, 1, 7)])
def test_encode_other_geometries(oracle, ref, k, m, bb):
    orig = np.random.RandomState(k * 1000 + m).randint(0, 256, (k, bb)).astype(np.uint8)
    assert np.array_equal(oracle.cm256_encode(orig, m), ref_encode(ref, orig, m))


def _erasure_sets():
    rs = np.random.RandomState(11)
    yield "config4_fixed", set(range(1, 121, 5))                       # SURVEY 8d pattern A: 24 originals
    yield "config4_random", set(rs.choice(160, 24, replace=False).tolist()) | {0}
    yield "worst_32", set(rs.choice(128, 32, replace=False).tolist())
    yield "one_original_row128_first", {7}                               # RecoveryCount == 1 -> DecodeM1
    yield "one_original_row129", {7, 128}                                # the RecoveryCount == 1 quirk
    yield "recovery_only", {130, 140, 159}


@pytest.mark.parametrize("name,lost", list(_erasure_sets()))
def test_decode_matches(oracle, ref, name, lost):
    R = 32
    orig = signals.noise(128 * 127, 900).view(np.uint8).reshape(128, 508)
    allb = np.concatenate([orig, ref_encode(ref, orig, R)])
    got = [i for i in range(160) if i not in lost][:128]
    n_rec = sum(1 for i in got if i >= 128)
    if n_rec == 0:
        pytest.skip("nothing to decode")
    data_ref = np.ascontiguousarray(allb[got])
    idx_ref = np.array(got, np.uint8)
    rc_ref = ref.sdrref_cm256_decode(128, n_rec, 508, data_ref.ctypes.data, idx_ref.ctypes.data)
    data_orc = np.ascontiguousarray(allb[got])
    rc_orc, idx_orc = oracle.cm256_decode(data_orc, np.array(got, np.uint8), 128, n_rec)
    assert rc_orc == rc_ref
    assert np.array_equal(idx_orc, idx_ref)
    assert np.array_equal(data_orc, data_ref)
