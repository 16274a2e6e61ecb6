"""GPU parity of the matrix-core interpolator cascade (interp_mfma.hip): forced through the C ABI with
interp_path=mfma (sdrhip_ctx_set_option) and short spans so that small inputs exercise many waves, the VALU head / tail segments and
the bank state hand-over.  Bit-exact against the oracle (itself pinned to the compiled reference)."""
import os

import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sdrdaemon_amd as sd

    assert sd.device_count() > 0, "GPU tests need a GPU and libsdrhip.so"
    return sd.Context(0)


@pytest.fixture()
def mfma_path(ctx):
    import sdrdaemon_amd as sd

    try:
        ctx.set_option("interp_path", "mfma")
    except sd.SdrHipError:
        pytest.skip("K5m (matrix-core interpolator experiment) is not in the product library: make -C sdrdaemon_amd/csrc WITH_K5M=1")

    def span(n):
        ctx.set_option("interp_span", n)

    span(64)
    yield span
    ctx.set_option("interp_path", "valu")
    ctx.set_option("interp_span", 0)


@pytest.mark.parametrize("signal", sorted(signals.ALL))
def test_mfma_vs_oracle_all_signals(ctx, oracle, mfma_path, signal):
    """All stress signals, interpolation by 4 / 8 / 16, two ragged calls (state carried through the VALU segments)."""
    import sdrdaemon_amd as sd

    x = signals.ALL[signal](30000 + 77)
    for log2 in (4, 3, 2):
        d, od = sd.Interpolators(ctx, 1), oracle.interpolators()
        for seg in (x[:20001], x[20001:]):
            a = d.interpolate(log2, seg)
            b = od.interpolate(log2, seg)
            assert np.array_equal(a, b), (signal, log2, np.argwhere(a != b)[:4])


@pytest.mark.parametrize("span", [64, 128, 320, 4096])
def test_mfma_span_lengths(ctx, oracle, mfma_path, span):
    import sdrdaemon_amd as sd

    mfma_path(span)
    x = signals.noise(90000 + 13, 5)
    d, od = sd.Interpolators(ctx, 1), oracle.interpolators()
    pos = 0
    for c in (40000, 20016, 30000 - 3):
        a = d.interpolate(4, x[pos:pos + c])
        b = od.interpolate(4, x[pos:pos + c])
        pos += c
        assert np.array_equal(a, b), (span, c, np.argwhere(a != b)[:4])


def test_mfma_stream_bank_equals_valu(ctx, mfma_path):
    """Three independent streams in one launch (device memory, strided rows), default span planning, against the VALU
    kernel on 2^20 inputs per stream."""
    import torch

    import sdrdaemon_amd as sd

    mfma_path(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randint(-32768, 32768, (3, (1 << 20) + 40, 2), generator=g, device=dev, dtype=torch.int16)
    for log2 in (2, 3, 4):
        ctx.set_option("interp_path", "mfma")
        a = sd.Interpolators(ctx, 3).interpolate(log2, x)
        ctx.set_option("interp_path", "valu")
        b = sd.Interpolators(ctx, 3).interpolate(log2, x)
        ctx.synchronize()
        assert torch.equal(a, b), log2
