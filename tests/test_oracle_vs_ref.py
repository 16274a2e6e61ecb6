"""Pins the oracle's DSP half: the C restatement (oracle/sdr_oracle.c) must equal the REAL
reference classes (oracle/_ref, compiled from /root/reference by oracle/Makefile) on every
Decimators / Interpolators entry point, both rounding flavours (EO1 = USE_SSE4_1, DB),
sampleSize 8/12/16, ragged chunked calls with state carried across calls."""
import numpy as np
import pytest

import signals
from oracle_lib import Reference

pytestmark = pytest.mark.skipif(not Reference.available("eo1"), reason="oracle/_ref not built")

CHUNKS = (4096, 64, 1000, 6000, 130, 8000)


@pytest.mark.parametrize("flavour", ["eo1", "db"])
@pytest.mark.parametrize("signal", sorted(signals.ALL))
def test_decimators_all_entry_points(oracle, flavour, signal):
    ref = Reference(flavour)
    x = signals.ALL[signal](20000 + 37)
    for ss0 in (8, 12, 16):
        xs = (x >> (16 - ss0)).astype(np.int16)
        for fcpos in (0, 1, 2):
            for log2 in range(0, 7):
                od, rd = oracle.decimators(ref.bias), ref.decimators()
                pos = 0
                for chunk in CHUNKS:
                    seg = xs[pos:pos + chunk]
                    pos += chunk
                    if len(seg) < (1 << log2):
                        continue  # the reference's unsigned loop bound wraps: undefined there
                    a, sa = od.decimate(log2, fcpos, ss0, seg)
                    b, sb = rd.decimate(log2, fcpos, ss0, seg)
                    assert sa == sb, (ss0, fcpos, log2, chunk)
                    assert np.array_equal(a, b), (ss0, fcpos, log2, chunk)


@pytest.mark.parametrize("flavour", ["eo1", "db"])
def test_decimators_mode_switch_keeps_filter_state(oracle, flavour):
    """m_decimator2..64 are shared by every decimateN_* (Decimators.h:56-70): switching the
    mode between calls must see the other mode's history."""
    ref = Reference(flavour)
    x = signals.noise(6 * 4096, 99)
    od, rd = oracle.decimators(ref.bias), ref.decimators()
    plan = [(4, 2), (4, 0), (3, 1), (6, 2), (2, 2), (5, 0), (4, 2)]
    for i, (log2, fcpos) in enumerate(plan[:6]):
        seg = x[i * 4096:(i + 1) * 4096]
        a, _ = od.decimate(log2, fcpos, 16, seg)
        b, _ = rd.decimate(log2, fcpos, 16, seg)
        assert np.array_equal(a, b), (i, log2, fcpos)


@pytest.mark.parametrize("flavour", ["eo1", "db"])
@pytest.mark.parametrize("signal", sorted(signals.ALL))
def test_interpolators_all_entry_points(oracle, flavour, signal):
    ref = Reference(flavour)
    x = signals.ALL[signal](3000)
    for log2 in range(0, 7):
        oi, ri = oracle.interpolators(), ref.interpolators()
        pos = 0
        for chunk in (1000, 1, 17, 982, 1000):
            seg = x[pos:pos + chunk]
            pos += chunk
            assert np.array_equal(oi.interpolate(log2, seg), ri.interpolate(log2, seg)), (log2, chunk)


def test_interpolate64_reference_quirk(oracle):
    """interpolate64_cen emits 32 interpolated + 32 zero samples per input (Interpolators.cpp:363-606)."""
    x = signals.noise(64, 5)
    y64 = oracle.interpolators().interpolate(6, x).reshape(64, 64, 2)
    y32 = oracle.interpolators().interpolate(5, x).reshape(64, 32, 2)
    assert np.array_equal(y64[:, :32], y32)
    assert not y64[:, 32:].any()
