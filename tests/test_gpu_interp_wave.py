"""GPU parity of K5w, the barrier-free wave-private interpolator cascade (interp_wave.h), forced through the C ABI with
interp_path = wave and, where stated, short segments (interp_span) so that small inputs run many waves, the 64-input warm-up of
every segment and the bank-state hand-over; and K5 (interp_path = valu) held to the same expectations.  Bit-exact against the
reference goldens and the oracle (itself pinned to the compiled reference)."""
import hashlib

import numpy as np
import pytest

import signals
from golden_util import Golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sdrdaemon_amd as sd

    assert sd.device_count() > 0, "GPU tests need a GPU and libsdrhip.so"
    return sd.Context(0)


@pytest.fixture(params=["wave", "valu"])
def path(ctx, request):
    ctx.set_option("interp_path", request.param)

    def span(n):
        ctx.set_option("interp_span", n)

    yield span
    ctx.set_option("interp_path", "auto")
    ctx.set_option("interp_span", 0)


def test_reference_goldens(ctx, path):
    """every interpolateN_cen golden of the compiled reference (both flavours, ragged 256 / 1 / 255 / 512 calls)"""
    import sdrdaemon_amd as sd

    G = Golden()
    n = 0
    for case in G.cases:
        if case["kind"] != "interpolate":
            continue
        x = G.input(case)
        u = sd.Interpolators(ctx, 1)
        outs, pos = [], 0
        for c in case["chunks"]:
            outs.append(u.interpolate(case["log2"], x[pos:pos + c]))
            pos += c
        assert np.array_equal(np.concatenate(outs), G.expected(case)), case["key"]
        n += 1
    assert n == 2 * 3 * 7
    for b in G.big:
        if b["kind"] == "interpolate16_cen" and b["flavour"] == "eo1":
            x = signals.noise(1 << 20, b["seed"])[:b["n"]]
            y = sd.Interpolators(ctx, 1).interpolate(4, x)
            assert hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest() == b["sha256"]


@pytest.mark.parametrize("signal", sorted(signals.ALL))
def test_all_signals_all_ratios_ragged(ctx, oracle, path, signal):
    """all stress signals, x4 .. x64, three ragged calls (odd lengths: the packed stage-0 plane shifts by an odd count)"""
    import sdrdaemon_amd as sd

    x = signals.ALL[signal](9000 + 77)
    for log2 in (2, 3, 4, 5, 6):
        d, od = sd.Interpolators(ctx, 1), oracle.interpolators()
        pos = 0
        for c in (4001, 1, 2, 127, 129, 3000, 1817):
            a = d.interpolate(log2, x[pos:pos + c])
            b = od.interpolate(log2, x[pos:pos + c])
            pos += c
            assert np.array_equal(a, b), (signal, log2, c, np.argwhere(a != b)[:4])


@pytest.mark.parametrize("span", [128, 256, 384, 1024, 4096])
def test_segment_lengths(ctx, oracle, path, span):
    """forced segment lengths: many one-wave workgroups per stream, each warming up on the 64 inputs in front of its slice"""
    import sdrdaemon_amd as sd

    path(span)
    x = signals.noise(90000 + 13, 5)
    for log2 in (2, 4, 5):
        d, od = sd.Interpolators(ctx, 1), oracle.interpolators()
        pos = 0
        for c in (40000, 20016, 30000 - 3):
            a = d.interpolate(log2, x[pos:pos + c])
            b = od.interpolate(log2, x[pos:pos + c])
            pos += c
            assert np.array_equal(a, b), (span, log2, c, np.argwhere(a != b)[:4])


def test_stream_bank_device_memory(ctx, oracle, path):
    """five independent streams in one launch (device memory, strided rows, the planner's own segments), 2^18 + 40 inputs per
    stream, x4 / x16 / x64 against the oracle; then a second call on the same handles (bank state)"""
    import torch

    import sdrdaemon_amd as sd

    S, n = 5, (1 << 18) + 40
    x = np.stack([signals.noise(2 * n, 300 + s) for s in range(S)])
    xd = torch.from_numpy(x).cuda()
    for log2 in (2, 4, 6):
        u = sd.Interpolators(ctx, S)
        ous = [oracle.interpolators() for _ in range(S)]
        for part in (slice(0, n), slice(n, 2 * n)):
            y = u.interpolate(log2, xd[:, part])
            ctx.synchronize()
            y = y.cpu().numpy()
            for s in range(S):
                assert np.array_equal(y[s], ous[s].interpolate(log2, x[s, part])), (log2, s)


def test_four_wave_workgroups_ragged_tail(ctx, oracle, path):
    """launches of >= 4096 segments run K5w as workgroups of four independent waves; a segment count that is not a multiple of
    four (1370 per stream, three streams, 128-input segments) leaves the last workgroup two idle waves"""
    import torch

    import sdrdaemon_amd as sd

    path(128)
    S, n = 3, 128 * 1369 + 5
    x = np.stack([signals.noise(n, 900 + s) for s in range(S)])
    xp = torch.zeros((S, n + 3, 2), dtype=torch.int16, device="cuda")  # (device rows: strides are multiples of four samples)
    xp[:, :n] = torch.from_numpy(x).cuda()
    xd = xp[:, :n]
    for log2 in (2, 5):
        u = sd.Interpolators(ctx, S)
        y = u.interpolate(log2, xd)
        ctx.synchronize()
        y = y.cpu().numpy()
        for s in range(S):
            assert np.array_equal(y[s], oracle.interpolators().interpolate(log2, x[s])), (log2, s)
