"""The BENCHMARKED launches under the parity suite: exactly bench.py's calls (Decimators.decimate and
RxPipe.process_view on device memory, the planner's own choice of kernel and spans) on bench.py's sizes, compared as WHOLE
outputs with SHA-256 digests made from the compiled reference decimator (tests/golden/make_golden.py: headline_golden).
Inputs come from the counter-based generator signals.hash_noise (numpy there, its torch twin here)."""
import hashlib

import numpy as np
import pytest

import signals
from golden_util import Golden, headline

pytestmark = pytest.mark.gpu

H = headline()


@pytest.fixture(scope="module")
def ctx():
    import sdrdaemon_amd as sd

    assert sd.device_count() > 0, "GPU tests need a GPU and libsdrhip.so"
    return sd.Context(0)


@pytest.fixture(params=[1, 3], ids=["fused-launch", "two-streams"])
def plumbing(request, ctx):
    """where a pipelined pipe runs the waiting encode: inside the next decimator launch (rx_fused_kernel) or on the context's
    second stream beside it (overlap mode: LDS-DMA ring of depth 3, encoder workgroups co-resident)"""
    ctx.set_option("rx_fused", request.param)
    yield request.param
    ctx.set_option("rx_fused", 1)


def _bank(name):
    import torch

    b = H[name]
    return torch.stack([signals.hash_noise_torch(1 << b["log2n"], s, "cuda") for s in b["seeds"]]), b


def _sha(t):
    return hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()


def _rx(ctx, S):
    import sdrdaemon_amd as sd

    m = H["meta"]
    return sd.RxPipe(ctx, S, log2decim=4, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=m["nb_fec"],
                     center_frequency_khz=m["center_frequency_khz"], sample_rate=m["sample_rate"])


def _check_frames(view, b):
    fr = view.torch()
    for s in range(fr.shape[0]):
        assert fr.shape[1] == b["nframes"][s]
        assert _sha(fr[s]) == b["frames_sha256"][s], ("frames of stream", s)


def test_headline_bank_8_x_2p25(ctx):
    """bench.py's step and its configs[1] line: 8 streams x 2^25 samples; the planner must pick the matrix-core kernel with
    one wave per SIMD (124 waves per stream, spans of 33 792 samples) on an MI355X"""
    import torch

    import sdrdaemon_amd as sd

    x, b = _bank("bank8")
    S, n = x.shape[0], x.shape[1]
    d = sd.Decimators(ctx, S, sd.HB_EO1)
    y = torch.empty((S, n >> 4, 2), dtype=torch.int16, device=x.device)
    d.decimate(4, sd.FC_CEN, 16, x, out=y)
    ctx.synchronize()
    plan = d.last_plan()
    assert plan["path"] == "mfma" and plan["wps"] == 124 and plan["span"] == 33792, plan
    for s in range(S):
        assert _sha(y[s]) == b["dec_sha256"][s], ("decimated stream", s)
    # a second call on the same handles continues the streams: same input again = a different output (history), still exact
    # against the reference is checked by the chunked goldens; here: the pipe, fresh handles, whole frame stream
    rx = _rx(ctx, S)
    view = rx.process_view(x, tv_sec=H["meta"]["tv_sec"], tv_usec=H["meta"]["tv_usec"])
    ctx.synchronize()
    plan = rx.last_plan()
    assert plan["path"] == "mfma" and plan["wps"] == 124 and plan["span"] == 33792, plan
    _check_frames(view, b)


def test_headline_bank_on_testsource_input(ctx):
    """bench.py's `configs` line on the input every BASELINE config names (VERDICT r5 #3a): the TestSource bank's 8 streams x 2^25
    samples (tests/headline_inputs.py) generated on the device -- their digests against the oracle NCO's samples --, through the
    decimator and through the Rx pipe: whole outputs against the digests the compiled reference made of the oracle's samples in
    the build container (headline_golden.json ts_bank8)"""
    import torch

    import headline_inputs as hi
    import sdrdaemon_amd as sd

    b = H["ts_bank8"]
    S, n = len(b["seeds"]), 1 << b["log2n"]
    ts = sd.TestSource(ctx, S)
    for s, seed in enumerate(b["seeds"]):
        assert hi.ts_config_string(seed) == b["config"][s]
        assert ts.configure(b["config"][s], s), ts.error()
    x = ts.read(n).reshape(S, n, 2).contiguous()
    ctx.synchronize()
    for s in (0, S - 1):
        assert _sha(x[s]) == b["input_sha256"][s], ("TestSource samples of stream", s)
    d = sd.Decimators(ctx, S, sd.HB_EO1)
    y = torch.empty((S, n >> 4, 2), dtype=torch.int16, device=x.device)
    d.decimate(4, sd.FC_CEN, 16, x, out=y)
    ctx.synchronize()
    assert d.last_plan()["path"] == "mfma"
    for s in range(S):
        assert _sha(y[s]) == b["dec_sha256"][s], ("decimated stream", s)
    del y
    rx = _rx(ctx, S)
    view = rx.process_view(x, tv_sec=H["meta"]["tv_sec"], tv_usec=H["meta"]["tv_usec"])
    ctx.synchronize()
    assert rx.last_plan()["path"] == "mfma"
    _check_frames(view, b)


def test_headline_bank_ring_depth_3(ctx):
    """the LDS-DMA ring of depth 3 (108 KiB per workgroup: what the decimator runs beside the encoder in overlap mode), on its
    own: the reference digests of the 8 x 2^25 bank"""
    import torch

    import sdrdaemon_amd as sd

    x, b = _bank("bank8")
    S, n = x.shape[0], x.shape[1]
    ctx.set_option("mfma_ring", 3)
    try:
        d = sd.Decimators(ctx, S, sd.HB_EO1)
        y = torch.empty((S, n >> 4, 2), dtype=torch.int16, device=x.device)
        d.decimate(4, sd.FC_CEN, 16, x, out=y)
        ctx.synchronize()
        assert d.last_plan()["path"] == "mfma"
        for s in range(S):
            assert _sha(y[s]) == b["dec_sha256"][s], ("decimated stream", s)
    finally:
        ctx.set_option("mfma_ring", 4)


@pytest.mark.parametrize("fused", [0, 1])
def test_headline_rx_fused_and_separate_launches(ctx, fused):
    """both plumbing variants of the Rx step (encoder inside the decimator's launch / separate launches): same frames"""
    x, b = _bank("bank8")
    ctx.set_option("rx_fused", fused)
    try:
        rx = _rx(ctx, x.shape[0])
        view = rx.process_view(x, tv_sec=H["meta"]["tv_sec"], tv_usec=H["meta"]["tv_usec"])
        ctx.synchronize()
        _check_frames(view, b)
    finally:
        ctx.set_option("rx_fused", 1)


@pytest.mark.parametrize("direct,enc", [(0, "fft"), (1, "karatsuba"), (0, "karatsuba"), (1, "fft")])
def test_headline_rx_arrangements(ctx, direct, enc):
    """the Rx step's data paths (round 5): the matrix-core decimator storing straight into the frame layout (rx_direct 1) or in
    stream order with K2 + the encoder's fused copy behind it (0), the CM256 encoder as additive FFT or as Karatsuba walk: the same
    frames, bit for bit, in every combination -- two calls, so that the second one begins inside an open frame"""
    import torch

    x, b = _bank("bank8")
    ctx.set_option("rx_direct", direct)
    ctx.set_option("enc_path", enc)
    try:
        rx = _rx(ctx, x.shape[0])
        view = rx.process_view(x, tv_sec=H["meta"]["tv_sec"], tv_usec=H["meta"]["tv_usec"])
        ctx.synchronize()
        _check_frames(view, b)
        # a second, ragged pair of calls against one call of another pipe in the default arrangement
        n1 = 16 * 70001
        ra = _rx(ctx, x.shape[0])
        fa = [ra.process_view(x[:, :n1], 1, 2).torch().clone(), ra.process_view(x[:, n1:2 * n1], 3, 4).torch().clone()]
        ctx.set_option("rx_direct", 1); ctx.set_option("enc_path", "fft")
        rb = _rx(ctx, x.shape[0])
        fb = [rb.process_view(x[:, :n1], 1, 2).torch().clone(), rb.process_view(x[:, n1:2 * n1], 3, 4).torch().clone()]
        ctx.synchronize()
        for u, v in zip(fa, fb):
            assert u.shape == v.shape and torch.equal(u, v)
    finally:
        ctx.set_option("rx_direct", 1)
        ctx.set_option("enc_path", "fft")


@pytest.mark.parametrize("log2,S,n,R,bias", [(2, 3, 1 << 21, 8, 0), (3, 2, 1 << 22, 1, 1), (4, 2, 1 << 23, 32, 0), (5, 8, 1 << 22, 32, 0),
                                             (6, 4, 1 << 23, 16, 1), (6, 1, 1 << 25, 32, 0)])
def test_rx_direct_framing_every_matrix_core_cascade(ctx, oracle, log2, S, n, R, bias):
    """the matrix-core decimator's frame-layout stores (rx_direct) for decimate4 .. 64_cen, both filter flavours: three ragged calls
    -- the second and third begin inside an open frame, blocks and frames end inside a lane pair's two samples -- against the ORACLE
    chain on the same calls (decimator -> UDPSinkFEC::write framer -> frame_encode, VERDICT r5 #7: no self-comparison), and against
    the stream-order arrangement (K2 + fused copy) of the same library; every call must have run the matrix-core kernel"""
    import numpy as np
    import torch

    import sdrdaemon_amd as sd
    import signals

    x = torch.stack([signals.hash_noise_torch(n, 7000 + 13 * log2 + s, "cuda") for s in range(S)])
    cuts = [0, (n // 3) & ~((1 << log2) * 4 - 1), (2 * n // 3 + 4096) & ~((1 << log2) * 4 - 1), n]
    stamps = [(5, 6), (77, 999999), (1000, 0)]
    ctx.set_option("decim_path", "mfma")
    try:
        outs = []
        for direct in (0, 1):
            ctx.set_option("rx_direct", direct)
            rx = sd.RxPipe(ctx, S, log2decim=log2, fcpos=sd.FC_CEN, hb_variant=bias, sample_bits=16, nb_fec=R,
                           center_frequency_khz=435000, sample_rate=48000)
            fr = []
            for (a, b), (ts, tu) in zip(zip(cuts[:-1], cuts[1:]), stamps):
                fr.append(rx.process_view(x[:, a:b], tv_sec=ts, tv_usec=tu).torch().clone())
                assert rx.last_plan()["path"] == "mfma", (direct, rx.last_plan())
            outs.append(fr)
        ctx.synchronize()
        for u, v in zip(*outs):
            assert u.shape == v.shape and u.shape[1] > 0 and torch.equal(u, v)
        got = [f.cpu().numpy() for f in outs[1]]
    finally:
        ctx.set_option("rx_direct", 1)
        ctx.set_option("decim_path", "auto")
    xh = x.cpu().numpy()
    for s in range(S):
        od = oracle.decimators(bias)
        fr = oracle.framer(nb_fec_blocks=R, sample_rate=48000, sample_bytes=2, sample_bits=16)
        for i, ((a, b), (ts, tu)) in enumerate(zip(zip(cuts[:-1], cuts[1:]), stamps)):
            y, ss = od.decimate(log2, sd.FC_CEN, 16, xh[s, a:b])
            assert ss == 16
            fr.s.tv_sec, fr.s.tv_usec = ts, tu
            e = fr.write(y)
            g = got[i][s]
            assert g.shape[0] == e.shape[0], (log2, s, i, g.shape, e.shape)
            for f in range(e.shape[0]):
                assert np.array_equal(g[f, :128], e[f]), (log2, s, i, f)
                assert np.array_equal(g[f, 128:], oracle.frame_encode(e[f], R)), (log2, s, i, f)


def test_headline_one_stream_2p27(ctx):
    """configs[2] literally: one stream of 2^27 samples through the decimator and through the Rx pipe, whole outputs"""
    import torch

    import sdrdaemon_amd as sd

    x, b = _bank("one27")
    d = sd.Decimators(ctx, 1, sd.HB_EO1)
    y = torch.empty((1, x.shape[1] >> 4, 2), dtype=torch.int16, device=x.device)
    d.decimate(4, sd.FC_CEN, 16, x, out=y)
    ctx.synchronize()
    assert d.last_plan()["path"] == "mfma"
    assert _sha(y[0]) == b["dec_sha256"][0]
    del y
    rx = _rx(ctx, 1)
    view = rx.process_view(x, tv_sec=H["meta"]["tv_sec"], tv_usec=H["meta"]["tv_usec"])
    ctx.synchronize()
    _check_frames(view, b)


def test_headline_bank_64(ctx):
    """config 5's bank on one GPU (bench.py --streams 64 at 2^22 samples per stream): 64 streams, whole outputs"""
    import torch

    import sdrdaemon_amd as sd

    x, b = _bank("bank64")
    S, n = x.shape[0], x.shape[1]
    d = sd.Decimators(ctx, S, sd.HB_EO1)
    y = torch.empty((S, n >> 4, 2), dtype=torch.int16, device=x.device)
    d.decimate(4, sd.FC_CEN, 16, x, out=y)
    ctx.synchronize()
    assert d.last_plan()["path"] == "mfma"
    for s in range(S):
        assert _sha(y[s]) == b["dec_sha256"][s], s
    rx = _rx(ctx, S)
    view = rx.process_view(x, tv_sec=H["meta"]["tv_sec"], tv_usec=H["meta"]["tv_usec"])
    ctx.synchronize()
    _check_frames(view, b)


@pytest.mark.parametrize("path", ["valu", "mfma"])
def test_reference_goldens_through_both_kernels(ctx, path):
    """Reference goldens of the centred cascades through BOTH kernels.  (a) the 16384-sample goldens of decimate4 / 8 / 16_cen
    in ONE call (their chunks are multiples of 16, so the reference's chunked output is the single-call output) with spans of
    256 << (L - 2): the matrix-core kernel engages (asserted); (b) the 65536-sample goldens of decimate4 .. 64_cen, one call
    and a ragged two-call split (state handed over through the VALU head / tail pieces)."""
    import sdrdaemon_amd as sd

    ctx.set_option("decim_path", path)
    try:
        G = Golden()
        n = 0
        for case in G.cases:
            if case["kind"] != "decimate" or case["fcpos"] != 2 or not 2 <= case["log2"] <= 4:
                continue
            ctx.set_option("mfma_span", 64 << case["log2"])
            d = sd.Decimators(ctx, 1, case["bias"])
            o, ss = d.decimate(case["log2"], 2, case["sample_size"], G.input(case))
            assert d.last_plan()["path"] == path, (case["key"], d.last_plan())
            assert ss == case["sample_size_out"] and np.array_equal(o, G.expected(case)), case["key"]
            n += 1
        assert n == 2 * 3 * 7
        GL = Golden("dsp_golden_long")
        for case in GL.cases:
            ctx.set_option("mfma_span", 64 << case["log2"])
            x = GL.input(case)
            d = sd.Decimators(ctx, 1, case["bias"])
            pos, outs, used = 0, [], []
            for c in case["chunks"]:
                o, ss = d.decimate(case["log2"], 2, 16, x[pos:pos + c])
                used.append(d.last_plan()["path"])
                outs.append(o)
                pos += c
            assert np.array_equal(np.concatenate(outs), GL.expected(case)), case["key"]
            if len(case["chunks"]) == 1:
                assert used == [path], (case["key"], used)
    finally:
        ctx.set_option("decim_path", "auto")
        ctx.set_option("mfma_span", 0)


def test_headline_pipelined_fused_launch(ctx, plumbing):
    """bench.py's pipelined step: call N's decimator launch carries the encoder workgroups of call N - 1's frames
    (rx_fused_kernel).  Two steps over the same bank + flush: step 1 delivers nothing, step 2 delivers step 1's frames =
    the reference digests; the flushed frames (second step: streams continue, filter history) equal the unpipelined pipe's."""
    import torch

    x, b = _bank("bank8")
    m = H["meta"]
    ref = _rx(ctx, x.shape[0])
    ref.process_view(x, tv_sec=m["tv_sec"], tv_usec=m["tv_usec"])
    second = ref.process_view(x, tv_sec=7, tv_usec=9).torch().clone()
    import sdrdaemon_amd as sd

    rx = sd.RxPipe(ctx, x.shape[0], log2decim=4, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=m["nb_fec"],
                   center_frequency_khz=m["center_frequency_khz"], sample_rate=m["sample_rate"], pipelined=True)
    v0 = rx.process_view(x, tv_sec=m["tv_sec"], tv_usec=m["tv_usec"])
    assert v0.shape[1] == 0
    v1 = rx.process_view(x, tv_sec=7, tv_usec=9)
    ctx.synchronize()
    assert rx.last_plan()["path"] == "mfma"
    _check_frames(v1, b)
    v2 = rx.flush_view().torch()
    ctx.synchronize()
    assert torch.equal(v2, second)
    assert rx.flush_view().shape[1] == 0


@pytest.mark.parametrize("cfg", [(4, 32, 1), (4, 32, 3), (3, 16, 2), (2, 32, 1), (4, 8, 1), (1, 32, 2), (5, 32, 1)])
def test_pipelined_equals_unpipelined_ragged(ctx, oracle, cfg, plumbing):
    """pipelined mode on ragged host calls (frames straddling calls, calls too short for the matrix cores, the generic encoder
    at nb_fec = 8, the VALU path at decimate2, decimate32 = no fused launch): same frames, one call later, flush at the end"""
    import sdrdaemon_amd as sd

    log2, R, S = cfg
    ctx.set_option("decim_path", "mfma")
    ctx.set_option("mfma_span", 64 << log2)
    try:
        n = (4 * 16129 + 3000) << log2
        xs = np.stack([signals.noise(n, 50 + s) for s in range(S)])
        cuts = [0, n // 3 + 64, n // 3 + 64 + (500 << log2), (2 * n // 3) & ~63, n]
        a = sd.RxPipe(ctx, S, log2decim=log2, nb_fec=R)
        p = sd.RxPipe(ctx, S, log2decim=log2, nb_fec=R, pipelined=True)
        exp, got = [], []
        for i in range(4):
            seg = xs[:, cuts[i]:cuts[i + 1]]
            exp.append(a.process(seg, 10 + i, 5 * i))
            got.append(p.process(seg, 10 + i, 5 * i))
        got.append(p.flush())
        assert got[0].shape[1] == 0
        exp, got = np.concatenate(exp, axis=1), np.concatenate(got, axis=1)
        assert exp.shape[1] == 4 and np.array_equal(exp, got), cfg
        # and against the oracle chain (decimator -> framer fed call by call with the same stamps -> encoder) for every stream
        for s in range(S):
            od, fr, pos, ofr = oracle.decimators(0), oracle.framer(nb_fec_blocks=R), 0, []
            for i in range(4):
                y, ss = od.decimate(log2, 2, 16, xs[s, cuts[i]:cuts[i + 1]])
                fr.s.tv_sec, fr.s.tv_usec = 10 + i, 5 * i
                ofr.append(fr.write(y))
            ofr = np.concatenate(ofr)
            assert ss == 16 and ofr.shape[0] == 4
            for f in range(4):
                assert np.array_equal(got[s, f, :128], ofr[f]), (cfg, s, f, "frame vs oracle")
                assert np.array_equal(got[s, f, 128:], oracle.frame_encode(ofr[f], R)), (cfg, s, f, "FEC vs oracle")
    finally:
        ctx.set_option("decim_path", "auto")
        ctx.set_option("mfma_span", 0)


def test_pipelined_reconfigure_needs_a_flush(ctx):
    """frames that wait for delivery carry the old frame size: a fecblk change is refused until they are flushed"""
    import sdrdaemon_amd as sd

    x = signals.noise(3 * 16129 * 16, 5)
    rx = sd.RxPipe(ctx, 1, log2decim=4, nb_fec=32, pipelined=True)
    assert rx.process(x[:2 * 16129 * 16], 1, 2).shape[0] == 0
    with pytest.raises(sd.SdrHipError):
        rx.reconfigure(nb_fec=8)
    assert rx.flush().shape[1] == 2
    rx.reconfigure(nb_fec=8)
    rx.process(x[2 * 16129 * 16:], 3, 4)
    out = rx.flush()
    assert out.shape[1:] == (1, 136, 512)


# ---------------------------------------------------------------- the benchmarked Tx launch
def _tx_input(ctx):
    import headline_inputs as hi

    x, _ = _bank("bank8")
    rxf, keep = hi.tx_received_frames(ctx, x, H["meta"])
    del x
    return rxf, keep


@pytest.mark.parametrize("max_rows,dec_path", [(32, "syndrome"), (128, "syndrome"), (128, "dense")])
def test_headline_tx_bank_8_x_128_frames(ctx, max_rows, dec_path):
    """bench.py's configs[3] step, exactly: tx.process(rxf) on 8 streams x 128 frames (config 3's frames of the bank8 run, a
    distinct random 24-of-160 loss pattern per frame), with bench.py's dec_max_rows = 32 (no fallback launches), with the default
    128, and through the dense decoder; whole outputs against digests made from the oracle's decode + the REFERENCE's
    interpolate16_cen (headline_golden.json: tx_bank8).  The decoder alone against the payload digests as well."""
    import sdrdaemon_amd as sd

    T = H["tx_bank8"]
    rxf, keep = _tx_input(ctx)
    S, F = rxf.shape[0], rxf.shape[1]
    assert (S, F) == (len(T["seeds"]), T["frames"]) and len({k.tobytes() for k in keep}) == S * F
    ctx.set_option("dec_max_rows", max_rows)
    ctx.set_option("dec_path", dec_path)
    before = ctx.counter("dec_rows_exceeded")
    try:
        tx = sd.TxPipe(ctx, S, T["log2interp"])
        iq = tx.process(rxf)
        ctx.synchronize()
        assert iq.shape == (S, (F * 16129) << T["log2interp"], 2)
        for s in range(S):
            assert _sha(iq[s]) == T["iq_sha256"][s], ("tx output of stream", s)
        del iq, tx
        pay = sd.fec_decode_frames(ctx, rxf.reshape(S * F, 128, 512))
        ctx.synchronize()
        pay = pay[0] if isinstance(pay, tuple) else pay
        pay = pay.reshape(S, F, -1)
        for s in range(S):
            assert _sha(pay[s]) == T["payload_sha256"][s], ("decoded payload of stream", s)
        assert ctx.counter("dec_rows_exceeded") == before
    finally:
        ctx.set_option("dec_max_rows", 128)
        ctx.set_option("dec_path", "syndrome")


@pytest.mark.parametrize("overlap", [1, 0], ids=["two-streams", "one-stream"])
def test_headline_tx_bank_pipelined(ctx, overlap):
    """bench.py's pipelined configs[3] step (sdrhip_tx_set_pipelined): call N decodes its batch on the second stream while the
    first one interpolates batch N - 1 and delivers it.  Three calls on the same received frames + flush: call 1 delivers nothing,
    call 2 delivers batch 1 = the reference digests (tx_bank8), the later deliveries equal the unpipelined pipe's second and third
    outputs (the interpolator histories continue)."""
    import torch

    import sdrdaemon_amd as sd

    T = H["tx_bank8"]
    rxf, _ = _tx_input(ctx)
    S, F = rxf.shape[0], rxf.shape[1]
    ctx.set_option("dec_max_rows", 32)
    ctx.set_option("tx_overlap", overlap)
    try:
        ref = sd.TxPipe(ctx, S, T["log2interp"])
        ref.process(rxf)
        second = ref.process(rxf).clone()
        third_sha = [_sha(v) for v in ref.process(rxf)]
        del ref
        tx = sd.TxPipe(ctx, S, T["log2interp"], pipelined=True)
        v0 = tx.process(rxf)
        assert v0.shape == (S, 0, 2)
        v1 = tx.process(rxf)
        ctx.synchronize()
        assert v1.shape == (S, (F * 16129) << T["log2interp"], 2)
        for s in range(S):
            assert _sha(v1[s]) == T["iq_sha256"][s], ("pipelined tx output of stream", s)
        del v1
        v2 = tx.process(rxf)
        ctx.synchronize()
        assert torch.equal(v2, second)
        del v2, second
        v3 = tx.flush(device=rxf.device)
        ctx.synchronize()
        assert [_sha(v) for v in v3] == third_sha
        assert tx.flush(device=rxf.device).shape == (S, 0, 2)
    finally:
        ctx.set_option("dec_max_rows", 128)
        ctx.set_option("tx_overlap", 1)


@pytest.mark.parametrize("overlap,device", [(1, False), (1, True), (0, False), (1, "pinned")],
                         ids=["two-streams-host", "two-streams-device", "one-stream-host", "two-streams-pinned-host"])
def test_tx_pipelined_many_ragged_calls(ctx, oracle, overlap, device):
    """20 pipelined Tx calls of 1..9 frames on two streams, a different random loss pattern per frame, the interpolation factor
    changed by control messages on the way (a waiting batch keeps the factor it was handed in with), an empty call in the middle
    (delivers like any other): every delivery equals the unpipelined pipe's output of the previous call, and the whole sample
    stream equals the oracle chain (decode -> interpolators with carried histories).  "pinned": the received frames come from
    sdrhip_host_alloc memory -- their upload on the second stream is then truly asynchronous -- and are scribbled over the moment
    the call returns (a host buffer belongs to the caller again when the call returns)."""
    import torch

    import sdrdaemon_amd as sd

    S, R = 2, 32
    rs = np.random.RandomState(23)
    counts = [int(v) for v in rs.choice([1, 2, 3, 5, 9], 20)]
    counts[6] = 0
    factors = [4] * 5 + [2] * 5 + [6] * 3 + [3] * 7
    Ftot = sum(counts)
    ys = [signals.noise(Ftot * 16129, 300 + s) for s in range(S)]
    rxb = np.zeros((S, Ftot, 128, 512), np.uint8)
    for s in range(S):
        frames = oracle.framer(nb_fec_blocks=R).write(ys[s])
        for f in range(Ftot):
            allb = np.concatenate([frames[f], oracle.frame_encode(frames[f], R)])
            lost = set(rs.choice(160, 24, replace=False).tolist())
            rxb[s, f] = allb[[i for i in range(160) if i not in lost][:128]]
    pinned = device == "pinned"
    device = device is True
    pin = ctx.host_alloc((S, 9, 128, 512), np.uint8) if pinned else None
    ctx.set_option("tx_overlap", overlap)
    try:
        a = sd.TxPipe(ctx, S, 4)
        p = sd.TxPipe(ctx, S, 4, pipelined=True)
        pos, prev, got_all = 0, None, []
        for i, c in enumerate(counts):
            assert a.configure({"interp": str(factors[i])}) and p.configure({"interp": str(factors[i])})
            batch = rxb[:, pos:pos + c]
            pos += c
            e = a.process(batch) if c else np.zeros((S, 0, 2), np.int16)
            if pinned and c:
                src = pin.reshape(-1)[:S * c * 128 * 512].reshape(S, c, 128, 512)  # (contiguous: passed to the library in place)
                src[:] = batch
                g = p.process(src)
                pin[:] = 0xA5  # the buffer is the caller's again
            else:
                g = p.process(torch.from_numpy(np.ascontiguousarray(batch)).cuda() if device else batch)
            g = g.cpu().numpy() if device else g
            if prev is None:
                assert g.shape == (S, 0, 2)
            else:
                assert g.shape == prev.shape and np.array_equal(g, prev), ("call", i)
            got_all.append(g)
            prev = e
        last = p.flush(device=torch.device("cuda", 0)).cpu().numpy() if device else p.flush()
        assert np.array_equal(last, prev)
        got_all.append(last)
        got = np.concatenate(got_all, axis=1)
        for s in range(S):
            ou, exp, q = oracle.interpolators(), [], 0
            for i, c in enumerate(counts):
                exp.append(ou.interpolate(factors[i], ys[s][q * 16129:(q + c) * 16129]))
                q += c
            assert np.array_equal(got[s], np.concatenate(exp)), ("stream", s)
        p.process(rxb[:, :1])
        assert ctx.lib.sdrhip_tx_set_pipelined(p.h, 0) != 0  # refused: a batch waits for delivery
        p.flush()
        assert ctx.lib.sdrhip_tx_set_pipelined(p.h, 0) == 0
    finally:
        ctx.set_option("tx_overlap", 1)
        if pin is not None:
            ctx.synchronize()
            ctx.host_free(pin)


@pytest.mark.parametrize("dec_path", ["syndrome", "dense"])
def test_dec_max_rows_is_checked_on_the_device(ctx, oracle, dec_path):
    """dec_max_rows is a promise; a frame that breaks it (33 recovery blocks under dec_max_rows = 32) is left as received --
    missing originals read zero, like an undecodable frame -- and COUNTED (sdrhip_ctx_get_counter), never half repaired and
    never silently; the same batch under dec_max_rows = 128 decodes completely."""
    import sdrdaemon_amd as sd

    rs = np.random.RandomState(5)
    frames = rs.randint(0, 256, (3, 128, 512)).astype(np.uint8)
    frames[:, :, 0:2] = 0
    frames[:, :, 2] = np.arange(128)
    frames[:, :, 3] = 0
    allb = np.stack([np.concatenate([frames[f], oracle.frame_encode(frames[f], 40)]) for f in range(3)])
    lost = [sorted(rs.choice(np.arange(1, 128), n, replace=False).tolist()) for n in (32, 33, 24)]  # frame 1 breaks the promise
    keeps = [[i for i in range(168) if i not in set(l)][:128] for l in lost]
    rx = np.stack([allb[f][keeps[f]] for f in range(3)])
    ctx.set_option("dec_path", dec_path)
    try:
        ctx.set_option("dec_max_rows", 32)
        c0 = ctx.counter("dec_rows_exceeded")
        pay, b0 = sd.fec_decode_frames(ctx, rx, want_block0=True)
        assert ctx.counter("dec_rows_exceeded") == c0 + 1
        for f in (0, 2):
            assert np.array_equal(pay[f].reshape(127, 508), frames[f, 1:, 4:]), f
        exp = frames[1, 1:, 4:].copy()
        exp[[i - 1 for i in lost[1]]] = 0
        assert np.array_equal(pay[1].reshape(127, 508), exp), "a frame beyond the promise keeps what was received, holes = 0"
        ctx.set_option("dec_max_rows", 128)
        pay, b0 = sd.fec_decode_frames(ctx, rx, want_block0=True)
        assert ctx.counter("dec_rows_exceeded") == c0 + 1
        for f in range(3):
            assert np.array_equal(pay[f].reshape(127, 508), frames[f, 1:, 4:]), f
    finally:
        ctx.set_option("dec_max_rows", 128)
        ctx.set_option("dec_path", "syndrome")


def test_pipelined_many_calls_wrap_the_frame_window(ctx, oracle, plumbing):
    """24 pipelined calls of varying size -- enough to exceed the frame area (cap_frames = 4 x (frames of the
    biggest call so far + 1)) several times, to wrap it while frames wait for delivery (wrap_hits_late -> a new area, the old
    one handed over as old_work) and to alternate the two stream-order buffers well beyond 4 calls; an empty call in the middle
    delivers like any other.  Every delivered frame against the unpipelined pipe AND the oracle chain."""
    import sdrdaemon_amd as sd

    log2, R, S = 4, 32, 2
    ctx.set_option("decim_path", "mfma")
    ctx.set_option("mfma_span", 64 << log2)
    try:
        rs = np.random.RandomState(17)
        sizes = [int(v) << log2 for v in rs.choice([700, 16129, 20000, 40000, 3 * 16129, 90000, 5000, 64516], 24)]
        sizes[7] = 0            # an empty call: delivers call 6's frames
        sizes[15] = 150000 << log2  # a big one: grows the area while call 14's frames wait
        n = sum(sizes)
        xs = np.stack([signals.noise(n, 80 + s) for s in range(S)])
        a = sd.RxPipe(ctx, S, log2decim=log2, nb_fec=R)
        p = sd.RxPipe(ctx, S, log2decim=log2, nb_fec=R, pipelined=True)
        pos, prev = 0, None
        got_all = []
        for i, c in enumerate(sizes):
            seg = xs[:, pos:pos + c]
            pos += c
            e = a.process(seg, 100 + i, 3 * i) if c else np.zeros((S, 0, 128 + R, 512), np.uint8)
            g = p.process(seg, 100 + i, 3 * i)
            if prev is not None:
                assert g.shape == prev.shape and np.array_equal(g, prev), ("call", i)
            else:
                assert g.shape[1] == 0
            got_all.append(g)
            prev = e
        last = p.flush()
        assert np.array_equal(last, prev)
        got_all.append(last)
        got = np.concatenate(got_all, axis=1)
        assert got.shape[1] == (n >> log2) // 16129
        for s in range(S):
            od, fr, ofr, q = oracle.decimators(0), oracle.framer(nb_fec_blocks=R), [], 0
            for i, c in enumerate(sizes):
                y, _ = od.decimate(log2, 2, 16, xs[s, q:q + c])
                q += c
                fr.s.tv_sec, fr.s.tv_usec = 100 + i, 3 * i
                ofr.append(fr.write(y))
            ofr = np.concatenate(ofr)
            assert ofr.shape[0] == got.shape[1]
            assert np.array_equal(got[s, :, :128], ofr), ("frames vs oracle", s)
            for f in range(0, ofr.shape[0], 7):
                assert np.array_equal(got[s, f, 128:], oracle.frame_encode(ofr[f], R)), (s, f)
    finally:
        ctx.set_option("decim_path", "auto")
        ctx.set_option("mfma_span", 0)
