"""GPU parity of the matrix-core decimator cascade (decim_mfma.hip): forced through the C ABI with
decim_path=mfma (sdrhip_ctx_set_option) and short spans so that small inputs exercise many waves, the VALU head / tail pieces
and the bank state hand-over.  Bit-exact against the oracle (itself pinned to the compiled reference)."""
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
    """forces the matrix-core cascade with short spans on the module's context (sdrhip_ctx_set_option)"""
    ctx.set_option("decim_path", "mfma")

    def span(n):
        ctx.set_option("mfma_span", n)

    span(1024)
    yield span
    ctx.set_option("decim_path", "auto")
    ctx.set_option("mfma_span", 0)


@pytest.mark.parametrize("signal", sorted(signals.ALL))
def test_mfma_vs_oracle_all_signals(ctx, oracle, mfma_path, signal):
    """All stress signals (full-scale alternation forces int32 wrap-around in the late stages), both rounding
    modes, centred decimation by 4 / 8 / 16, two ragged calls (state carried through the VALU pieces)."""
    import sdrdaemon_amd as sd

    x = signals.ALL[signal](300000 + 77)
    for bias in (0, 1):
        for log2 in (6, 5, 4, 3, 2):
            d, od = sd.Decimators(ctx, 1, bias), oracle.decimators(bias)
            for seg in (x[:200001], x[200001:]):
                a, sa = d.decimate(log2, 2, 16, seg)
                b, sb = od.decimate(log2, 2, 16, seg)
                assert sa == sb
                assert np.array_equal(a, b), (signal, bias, log2, np.argwhere(a != b)[:4])


@pytest.mark.parametrize("span", [1024, 2048, 5120, 16384])
def test_mfma_span_lengths_and_sample_sizes(ctx, oracle, mfma_path, span):
    import sdrdaemon_amd as sd

    mfma_path(span)
    x = signals.noise(700000 + 13, 5)
    for ss in (8, 12, 16):
        xs = (x >> (16 - ss)).astype(np.int16)
        d, od = sd.Decimators(ctx, 1, 0), oracle.decimators(0)
        pos = 0
        for c in (300000, 150016, 250000 - 3):
            a, sa = d.decimate(4, 2, ss, xs[pos:pos + c])
            b, sb = od.decimate(4, 2, ss, xs[pos:pos + c])
            pos += c
            assert sa == sb
            assert np.array_equal(a, b), (span, ss, c, np.argwhere(a != b)[:4])


def test_mfma_stream_bank(ctx, oracle, mfma_path):
    """Three independent streams in one launch (device memory, strided rows)."""
    import torch

    import sdrdaemon_amd as sd

    n = 262144 + 64
    xs = [signals.noise(n, 40 + s) for s in range(3)]
    d = sd.Decimators(ctx, 3, 0)
    xd = torch.from_numpy(np.stack(xs)).cuda()
    y, ss = d.decimate(4, 2, 16, xd)
    ctx.synchronize()
    y = y.cpu().numpy()
    for s in range(3):
        b, _ = oracle.decimators(0).decimate(4, 2, 16, xs[s])
        assert np.array_equal(y[s], b), s


@pytest.mark.parametrize("nstreams", [1, 3, 40])
def test_mfma_default_span_planning_equals_valu(ctx, mfma_path, nstreams):
    """The planner's own span choice (one wave per SIMD from decimate16 up, short spans for small calls, banks of many
    streams) against the VALU kernel, two calls each so that the bank state crosses paths."""
    import torch

    import sdrdaemon_amd as sd

    mfma_path(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3 + nstreams)
    for n in ((70000, 300000 + 12) if nstreams > 3 else (70000, 300000 + 12, (1 << 21) + 76)):
        x = torch.randint(-32768, 32768, (nstreams, n, 2), generator=g, device=dev, dtype=torch.int16)
        for log2 in (2, 4, 5, 6):
            res = []
            for path in ("mfma", "valu"):
                ctx.set_option("decim_path", path)
                d = sd.Decimators(ctx, nstreams, 0)
                cut = (n // 2 + 8) & ~3  # (device rows must stay 16-byte aligned)
                a, _ = d.decimate(log2, 2, 16, x[:, :cut])
                b, _ = d.decimate(log2, 2, 16, x[:, cut:])
                res.append((a.clone(), b.clone()))
            ctx.synchronize()
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (nstreams, n, log2)


@pytest.mark.parametrize("nb_fec", [8, 32])
def test_mfma_rx_pipe_frames(ctx, oracle, mfma_path, nb_fec):
    """Rx pipe behind the matrix-core decimator: stream-order output, then UDPSinkFEC::write's layout by the framing
    kernel (nb_fec = 8: generic encoder) or by the 128-original encoder itself (nb_fec = 32: only the frame open at
    either end of a call goes through the framing kernel); three ragged calls, frames spanning calls."""
    import sdrdaemon_amd as sd

    for span in (1024, 8192):
        mfma_path(span)
        nsamp = (5 * 16129 + 5000) << 4
        x = signals.noise(nsamp, 31, 16)
        rx = sd.RxPipe(ctx, 1, log2decim=4, fcpos=sd.FC_CEN, hb_variant=0, sample_bits=16, nb_fec=nb_fec,
                       center_frequency_khz=435000, sample_rate=625000)
        od = oracle.decimators(0)
        fr = oracle.framer(nb_fec_blocks=nb_fec)
        cuts = [0, nsamp // 3 + 80, nsamp // 3 + 80 + 16 * 1000, nsamp]
        got, exp = [], []
        for i in range(3):
            seg = x[cuts[i]:cuts[i + 1]]
            got.append(rx.process(seg, tv_sec=100 + i, tv_usec=7 * i))
            y, ss = od.decimate(4, 2, 16, seg)
            fr.s.tv_sec, fr.s.tv_usec = 100 + i, 7 * i
            exp.append(fr.write(y))
        got, exp = np.concatenate(got), np.concatenate(exp)
        assert got.shape[0] == exp.shape[0] == 5
        for f in range(5):
            assert np.array_equal(got[f, :128], exp[f]), (span, f)
            assert np.array_equal(got[f, 128:], oracle.frame_encode(exp[f], nb_fec)), (span, f)


def test_mfma_ring_depths_agree(ctx, oracle, mfma_path):
    """decimate16 on the LDS-DMA ring at every depth the library has -- 4 groups (the product: one wave per SIMD), 3 (beside another kernel),
    2 (the two-waves-per-SIMD experiment of round 6: spans half as long, a whole group issued per 8 steps) -- : the same bytes, through the
    decimator alone (stream 0 against the oracle) and through the Rx pipe's frame-layout stores (two ragged calls)."""
    import torch

    import sdrdaemon_amd as sd

    mfma_path(0)
    S, n = 8, (1 << 23) + 4 * 1111
    x = np.stack([signals.noise(n, 900 + s) for s in range(S)])
    xd = torch.from_numpy(x).cuda()
    exp0, _ = oracle.decimators(0).decimate(4, 2, 16, x[0])
    outs, frames, plans = [], [], []
    try:
        for ring in (4, 3, 2):
            ctx.set_option("mfma_ring", ring)
            d = sd.Decimators(ctx, S, 0)
            y, _ = d.decimate(4, 2, 16, xd)
            plans.append(d.last_plan())
            outs.append(y.clone())
            rx = sd.RxPipe(ctx, S, log2decim=4, nb_fec=32)
            cut = (n // 3) & ~3
            frames.append(torch.cat([rx.process(xd[:, :cut], 1, 2), rx.process(xd[:, cut:], 3, 4)], dim=1).clone())
        ctx.synchronize()
    finally:
        ctx.set_option("mfma_ring", 4)
    assert all(p["path"] == "mfma" for p in plans), plans
    assert plans[2]["wps"] > plans[0]["wps"], plans  # (ring 2 really planned two waves per SIMD)
    assert np.array_equal(outs[0][0].cpu().numpy(), exp0)
    for k in (1, 2):
        assert torch.equal(outs[k], outs[0]), k
        assert torch.equal(frames[k], frames[0]), k
