"""GPU parity of the CM256 path (through the C ABI) vs the oracle.  Bit-exact."""
import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sdrdaemon_amd as sd

    assert sd.device_count() > 0
    return sd.Context(0)


def test_cm256_encode_matches_oracle(ctx, oracle):
    import sdrdaemon_amd as sd

    cm = sd.CM256(ctx)
    rs = np.random.RandomState(4)
    for k, m, bb in ((128, 32, 508), (128, 1, 508), (128, 128, 508), (5, 3, 33), (16, 16, 64), (2, 2, 1400), (100, 20, 1021)):
        x = rs.randint(0, 256, size=(k, bb)).astype(np.uint8)
        rc, rec = cm.cm256_encode((k, m, bb), x)
        assert rc == 0
        assert np.array_equal(rec, oracle.cm256_encode(x, m)), (k, m, bb)


def _deliver(x, rec, erased, rows):
    k = x.shape[0]
    keep = [i for i in range(k) if i not in set(erased)]
    data = np.concatenate([x[keep], rec[rows]]).copy()
    idx = np.array(keep + [k + r for r in rows])
    return data, idx


@pytest.mark.parametrize("case", ["stride5_24", "first32", "last32", "random24", "one_row0", "m1_quirk", "none"])
def test_cm256_decode_matches_oracle(ctx, oracle, case):
    import sdrdaemon_amd as sd

    cm = sd.CM256(ctx)
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
    elif case == "m1_quirk":
        erased, rows = [9], [2]
    else:
        erased, rows = [], []
    d1, i1 = _deliver(x, rec, erased, rows)
    d2 = d1.copy()
    nrec = max(len(rows), 1)
    rc1, j1 = cm.cm256_decode((128, nrec, 508), d1, i1)
    rc2, j2 = oracle.cm256_decode(d2, i1, 128, nrec)
    assert rc1 == rc2 == 0
    assert np.array_equal(j1, j2)
    assert np.array_equal(d1, d2)
    if case not in ("m1_quirk",):
        out = np.zeros_like(x)
        out[j1] = d1
        assert np.array_equal(out, x)


def test_decode_duplicate_index_is_an_error(ctx):
    import sdrdaemon_amd as sd

    cm = sd.CM256(ctx)
    data = np.zeros((128, 508), np.uint8)
    idx = np.arange(128)
    idx[5] = 4
    idx[127] = 130
    rc, _ = cm.cm256_decode((128, 2, 508), data, idx)
    assert rc == -5


def test_frames_batch_encode_decode(ctx, oracle):
    """Batched frame API (the Rx encode / Tx decode call sites), host and device memory, several
    erasure patterns in one batch including block 0 and lost recovery blocks."""
    import torch

    import sdrdaemon_amd as sd

    F, R = 11, 32
    x = signals.noise(F * 16129, 77)
    fr = oracle.framer(nb_fec_blocks=R, tv_sec=5, tv_usec=6)
    frames = fr.write(x)
    assert frames.shape[0] == F
    exp_rec = np.stack([oracle.frame_encode(frames[f], R) for f in range(F)])
    rec_h = sd.fec_encode_frames(ctx, frames, R)
    assert np.array_equal(rec_h, exp_rec)
    rec_d = sd.fec_encode_frames(ctx, torch.from_numpy(frames).cuda(), R)
    ctx.synchronize()
    assert np.array_equal(rec_d.cpu().numpy(), exp_rec)

    rs = np.random.RandomState(8)
    rx = np.zeros((F, 128, 512), np.uint8)
    for f in range(F):
        allb = np.concatenate([frames[f], exp_rec[f]])
        if f % 3 == 0:
            lost = set(range(1, 121, 5))  # pattern A: 24 originals at stride 5
        elif f % 3 == 1:
            lost = set(rs.choice(160, 24, replace=False).tolist()) | {0}  # pattern B incl. block 0
        else:
            lost = set()
        keep = [i for i in range(160) if i not in lost][:128]
        rx[f] = allb[keep]
    payload, b0 = sd.fec_decode_frames(ctx, rx, want_block0=True)
    for f in range(F):
        assert np.array_equal(payload[f].view(np.int16).reshape(-1, 2), x[f * 16129:(f + 1) * 16129]), f
        assert np.array_equal(b0[f], frames[f, 0, 4:]), f
    pd = sd.fec_decode_frames(ctx, torch.from_numpy(rx).cuda())
    ctx.synchronize()
    assert np.array_equal(pd.cpu().numpy(), payload)


def test_incomplete_frame_keeps_zeros(ctx, oracle):
    """< 128 blocks cannot happen in the batched API (it takes the first 128), but a frame whose
    128 received blocks contain no recovery block is passed through untouched."""
    import sdrdaemon_amd as sd

    x = signals.noise(16129, 5)
    frames = oracle.framer(nb_fec_blocks=0).write(x)
    payload = sd.fec_decode_frames(ctx, frames)
    assert np.array_equal(payload[0].view(np.int16).reshape(-1, 2), x)


def test_decode_plan_cache_many_patterns(ctx, oracle):
    """Repeated batches with fresh random erasure patterns: the device-side plan cache (64 slots)
    fills, is reused and is recycled; every frame must still decode exactly."""
    import sdrdaemon_amd as sd

    R = 32
    rs = np.random.RandomState(123)
    x = signals.noise(30 * 16129, 55)
    frames = oracle.framer(nb_fec_blocks=R).write(x)
    rec = np.stack([oracle.frame_encode(f, R) for f in frames])
    fixed = sorted(rs.choice(160, 20, replace=False).tolist())
    for call in range(5):
        rx = np.zeros((30, 128, 512), np.uint8)
        for f in range(30):
            lost = set(fixed) if f % 2 == 0 else set(rs.choice(160, int(rs.randint(1, 33)), replace=False).tolist())
            allb = np.concatenate([frames[f], rec[f]])
            rx[f] = allb[[i for i in range(160) if i not in lost][:128]]
        payload, b0 = sd.fec_decode_frames(ctx, rx, want_block0=True)
        for f in range(30):
            assert np.array_equal(payload[f].view(np.int16).reshape(-1, 2), x[f * 16129:(f + 1) * 16129]), (call, f)
            assert np.array_equal(b0[f], frames[f, 0, 4:]), (call, f)


def test_decode_batch_with_more_patterns_than_cache_slots(ctx, oracle):
    """150 frames, every one with its own erasure pattern, in ONE call (round 1 pushed such a batch through a
    64-slot plan cache in chunks; the device planner takes it as it comes)."""
    import sdrdaemon_amd as sd

    R, F = 32, 150
    rs = np.random.RandomState(77)
    x = signals.noise(F * 16129, 56)
    frames = oracle.framer(nb_fec_blocks=R).write(x)
    rx = np.zeros((F, 128, 512), np.uint8)
    for f in range(F):
        lost = set(rs.choice(160, int(rs.randint(2, 33)), replace=False).tolist())
        if sum(1 for i in lost if i < 128) == 1:
            lost.discard(128)  # one lost original + lost row 128 = cm256's RecoveryCount == 1 shortcut quirk (test "m1_quirk")
        allb = np.concatenate([frames[f], oracle.frame_encode(frames[f], R)])
        rx[f] = allb[[i for i in range(160) if i not in lost][:128]]
    assert len({rx[f, :, 2].tobytes() for f in range(F)}) > 2 * 64
    payload, b0 = sd.fec_decode_frames(ctx, rx, want_block0=True)
    for f in range(F):
        assert np.array_equal(payload[f].view(np.int16).reshape(-1, 2), x[f * 16129:(f + 1) * 16129]), f
        assert np.array_equal(b0[f], frames[f, 0, 4:]), f


def test_device_resident_batch_plans_on_the_gpu(ctx, oracle):
    """Frames resident on the device, no index array from the host: the planner reads header.blockIndex itself.
    Every frame has its own loss pattern; up to 128 losses out of 256 blocks (every original may be gone:
    a 128 x 128 Cauchy block to invert), arrival order shuffled among the originals."""
    import torch

    import sdrdaemon_amd as sd

    R, F = 128, 24
    rs = np.random.RandomState(5)
    x = signals.noise(F * 16129, 91)
    frames = oracle.framer(nb_fec_blocks=127).write(x)
    frames[:, :, 3] = 0
    rx = np.zeros((F, 128, 512), np.uint8)
    for f in range(F):
        nlost = [128, 127, 100, 64, 33, 2, 0][f % 7] if f < 21 else int(rs.randint(2, 129))
        lost = set(rs.choice(256, nlost, replace=False).tolist())
        if sum(1 for i in lost if i < 128) == 1:
            lost.discard(128)  # (cm256's RecoveryCount == 1 shortcut quirk, covered by its own test)
        allb = np.concatenate([frames[f], oracle.frame_encode(frames[f], R)])
        got = [i for i in range(256) if i not in lost][:128]
        orig = [i for i in got if i < 128]
        rs.shuffle(orig)  # originals in any order, recovery blocks last (SDRdaemonFECBuffer.cpp:210)
        rx[f] = allb[orig + [i for i in got if i >= 128]]
    payload, b0 = sd.fec_decode_frames(ctx, torch.from_numpy(rx).cuda(), want_block0=True)
    ctx.synchronize()
    payload, b0 = payload.cpu().numpy(), b0.cpu().numpy()
    for f in range(F):
        assert np.array_equal(payload[f].view(np.int16).reshape(-1, 2), x[f * 16129:(f + 1) * 16129]), f
        assert np.array_equal(b0[f], frames[f, 0, 4:]), f


@pytest.mark.parametrize("R", [1, 12, 13, 16, 17, 31, 32, 33, 64, 127, 128])
def test_frame_encoder_every_row_count(ctx, oracle, R):
    """nb_fec below 13 uses the generic kernel, from 13 on the structured (Karatsuba over the
    XOR-convolution form of the Cauchy rows) one, in 16-row tiles: both must equal cm256_encode."""
    import sdrdaemon_amd as sd

    F = 5
    x = signals.noise(F * 16129, 200 + R)
    frames = oracle.framer(nb_fec_blocks=min(R, 127)).write(x)
    frames[:, :, 3] = 0
    rec = sd.fec_encode_frames(ctx, frames, R)
    for f in range(F):
        assert np.array_equal(rec[f], oracle.frame_encode(frames[f], R)), (R, f)


def test_syndrome_decoder_equals_dense_decoder_on_hostile_batches(ctx, oracle):
    """The batched decoder's two device paths -- syndrome kernel (encoder walk over the received originals + N x N inverse,
    default) and the dense N x 128 matrix kernel -- on frames that exercise every branch of the planner: 0 .. 40 erasures
    (> 32: the syndrome path hands the frame to the dense kernel), high recovery rows (sender fecblk 128), block 0 erased,
    cm256's one-recovery-block XOR shortcut on a row that is not the parity row, repeated originals and repeated recovery
    blocks (cm256's decode error: the frame keeps what was received), arrival order shuffled.  Both must give the same
    bytes; the decodable ones must give back the originals."""
    import sdrdaemon_amd as sd

    R, F = 128, 64
    rs = np.random.RandomState(2024)
    x = signals.noise(F * 16129, 77)
    frames = oracle.framer(nb_fec_blocks=127).write(x)
    frames[:, :, 3] = 0
    rx = np.zeros((F, 128, 512), np.uint8)
    decodable = np.ones(F, bool)
    for f in range(F):
        allb = np.concatenate([frames[f], oracle.frame_encode(frames[f], R)])
        kind = f % 8
        nlost = [0, 1, 3, 24, 32, 33, 40, 17][kind]
        lost_o = sorted(rs.choice(128, nlost, replace=False).tolist())
        if kind == 2:
            lost_o[0] = 0  # block 0 among the erased
            lost_o = sorted(set(lost_o))
        rows = sorted(rs.choice(128, len(lost_o), replace=False).tolist())  # any recovery rows, not the first ones
        if kind == 1:
            rows = [int(rs.randint(1, 128))]  # M1 shortcut with a non-parity row: cm256 XORs anyway (wrong bytes, mirrored)
            decodable[f] = False
        got = [i for i in range(128) if i not in lost_o]
        rs.shuffle(got)
        order = got + [128 + r for r in rows]
        if kind == 7 and f % 16 == 7:
            order[3] = order[4]  # a repeated original
            decodable[f] = False
        if kind == 7 and f % 16 == 15:
            order[-1] = order[-2]  # the same recovery block twice
            decodable[f] = False
        rx[f] = allb[order]
    outs = {}
    for path in ("dense", "syndrome"):
        ctx.set_option("dec_path", path)
        try:
            outs[path] = sd.fec_decode_frames(ctx, rx, want_block0=True)
        finally:
            ctx.set_option("dec_path", "syndrome")
    for k in (0, 1):
        assert np.array_equal(outs["dense"][k], outs["syndrome"][k]), ("payload", "block0")[k]
    payload, b0 = outs["syndrome"]
    for f in range(F):
        if decodable[f]:
            assert np.array_equal(payload[f].view(np.int16).reshape(-1, 2), x[f * 16129:(f + 1) * 16129]), f
            assert np.array_equal(b0[f], frames[f, 0, 4:]), f


@pytest.mark.parametrize("strict", [0, 1])
def test_fused_plan_equals_the_planning_kernel_on_hostile_batches(ctx, oracle, strict):
    """Round 6: with dec_max_rows <= 32 the FFT decoder's workgroups derive their frame's plan themselves (dec_plan = fused, one
    launch) instead of reading the record gf_decode_plan_kernel wrote (dec_plan = kernel).  Every planner branch a sender with
    fecblk <= 32 can cause -- 0 .. 32 erasures, block 0 erased, the one-recovery-block XOR shortcut on a non-parity row, a repeated
    original, a repeated recovery block, a frame with missing originals and too few recovery blocks (zero fill of what never
    came), a frame that breaks the dec_max_rows promise (left as received, counted), recovery blocks interleaved with originals
    in arrival order (strict mode's holes) -- must give the same bytes through both, and through the dense kernel; the decodable
    frames must give back the originals; device-resident indices and header-derived indices alike."""
    import sdrdaemon_amd as sd

    R, F = 32, 72
    rs = np.random.RandomState(606)
    x = signals.noise(F * 16129, 78)
    frames = oracle.framer(nb_fec_blocks=R).write(x)
    rx = np.zeros((F, 128, 512), np.uint8)
    decodable = np.ones(F, bool)
    for f in range(F):
        allb = np.concatenate([frames[f], oracle.frame_encode(frames[f], 64)[:64]])  # rows 0 .. 63 exist, a fecblk-32 sender uses 0 .. 31
        kind = f % 9
        nlost = [0, 1, 3, 24, 32, 17, 9, 5, 12][kind]
        lost_o = sorted(rs.choice(128, nlost, replace=False).tolist())
        if kind == 2:
            lost_o[0] = 0
            lost_o = sorted(set(lost_o))
        rows = sorted(rs.choice(32, len(lost_o), replace=False).tolist())
        if kind == 1:
            rows = [int(rs.randint(1, 32))]  # DecodeM1 with a non-parity row (wrong bytes, mirrored)
            decodable[f] = False
        got = [i for i in range(128) if i not in lost_o]
        rs.shuffle(got)
        order = got + [128 + r for r in rows]
        if kind == 5:
            rs.shuffle(order)  # recovery blocks anywhere in arrival order
            decodable[f] = not strict
        if kind == 6 and f % 18 == 6:
            order[3] = order[4]  # a repeated original
            decodable[f] = False
        if kind == 6 and f % 18 == 15:
            order[-1] = order[-2]  # the same recovery block twice
            decodable[f] = False
        if kind == 7:
            order[-2:] = order[:2]  # two recovery blocks replaced by repeats of originals: blocks missing for good (zero fill of what never came)
            decodable[f] = False
        if kind == 8:
            # 33 recovery blocks in a frame whose sender promised 32: left as received, counted
            order = [i for i in range(128) if i >= 33] + [128 + r for r in range(33)]
            decodable[f] = False
        assert len(order) == 128
        rx[f] = allb[order]
    outs, counts = {}, {}
    ctx.set_option("dec_max_rows", R)
    ctx.set_option("dec_strict", strict)
    try:
        for name, opts in (("fused", (("dec_plan", "fused"),)), ("kernel", (("dec_plan", "kernel"),)), ("dense", (("dec_path", "dense"),))):
            for k, v in opts:
                ctx.set_option(k, v)
            try:
                c0 = ctx.counter("dec_rows_exceeded")
                outs[name] = sd.fec_decode_frames(ctx, rx, want_block0=True)
                counts[name] = ctx.counter("dec_rows_exceeded") - c0
            finally:
                ctx.set_option("dec_plan", "fused")
                ctx.set_option("dec_path", "syndrome")
        idx = np.ascontiguousarray(rx[:, :, 2])
        outs["fused-indices"] = sd.fec_decode_frames(ctx, rx, indices=idx, want_block0=True)
        # the staggered start in both phase rules (experiment options: a sleep in front of the loads, the arrival-rank rule with a
        # barrier of its own): same bytes
        for mod in (104, 3):
            ctx.set_option("fec_stagger", 2)
            ctx.set_option("fec_stagger_mod", mod)
            try:
                outs["fused-stagger-%d" % mod] = sd.fec_decode_frames(ctx, rx, want_block0=True)
            finally:
                ctx.set_option("fec_stagger", 0)
                ctx.set_option("fec_stagger_mod", 0)
    finally:
        ctx.set_option("dec_max_rows", 128)
        ctx.set_option("dec_strict", 0)
    assert counts["fused"] == counts["kernel"] == counts["dense"] == F // 9
    for name in ("kernel", "dense", "fused-indices", "fused-stagger-104", "fused-stagger-3"):
        for k in (0, 1):
            bad = [f for f in range(F) if not np.array_equal(outs["fused"][k][f], outs[name][k][f])]
            assert not bad, (name, ("payload", "block0")[k], bad, [f % 9 for f in bad])
    payload, b0 = outs["fused"]
    for f in range(F):
        if decodable[f]:
            assert np.array_equal(payload[f].view(np.int16).reshape(-1, 2), x[f * 16129:(f + 1) * 16129]), (f, f % 9)
            assert np.array_equal(b0[f], frames[f, 0, 4:]), f


@pytest.mark.parametrize("dec_path", ["syndrome", "dense"])
def test_strict_mode_leaves_the_reference_s_holes(ctx, oracle, dec_path):
    """ctx option dec_strict (VERDICT r3 missing #4): the reference copies back only the descriptors [128 - recoveryCount, 128)
    after cm256_decode (SDRdaemonFECBuffer.cpp:204-211), so a block restored into a recovery block that arrived BEFORE some
    original stays a hole.  Frames with the recovery blocks interleaved among the originals in arrival order: with dec_strict = 1
    the batched decoder returns exactly what the oracle's restatement of SDRdaemonFECBuffer returns (holes and all), with the
    default every restored block is delivered (the whole frame)."""
    import sdrdaemon_amd as sd

    rs = np.random.RandomState(21)
    F, R = 12, 32
    x = signals.noise(F * 16129, 77)
    frames = oracle.framer(nb_fec_blocks=R).write(x)
    assert frames.shape[0] == F
    rx = np.zeros((F, 128, 512), np.uint8)
    exp_strict = np.zeros((F, 127 * 508), np.uint8)
    nholes = 0
    for f in range(F):
        allb = np.concatenate([frames[f], oracle.frame_encode(frames[f], R)])
        nlost = [0, 1, 5, 24, 32, 17, 2, 9, 31, 12, 3, 24][f]
        lost = set(rs.choice(np.arange(1, 128), nlost, replace=False).tolist())
        keep = [i for i in range(128 + R) if i not in lost][:128]
        if f % 3 != 0:
            keep = list(rs.permutation(keep))  # arrival order: recovery blocks anywhere among the originals
        if nlost == 1:  # (cm256's RecoveryCount == 1 shortcut only works with recovery row 128: keep that one)
            keep = [k for k in keep if k < 128] + [128]
        rx[f] = allb[keep]
        buf = oracle.fecbuffer()
        out = None
        for b in keep:
            assert buf.write_and_read(allb[b]) is None or True
        nxt = allb[0].copy()
        nxt[0] = (int(nxt[0]) + 1) & 0xff  # a block of the next frame closes this one
        out = buf.write_and_read(nxt)
        assert out is not None
        exp_strict[f] = out
        nholes += int((out.reshape(127, 508) != frames[f, 1:, 4:]).any(axis=1).sum())
    assert nholes > 0, "the test must contain frames the reference leaves holes in"
    ctx.set_option("dec_path", dec_path)
    try:
        ctx.set_option("dec_strict", 1)
        got = sd.fec_decode_frames(ctx, rx)
        assert np.array_equal(got, exp_strict)
        ctx.set_option("dec_strict", 0)
        got = sd.fec_decode_frames(ctx, rx)
        for f in range(F):
            assert np.array_equal(got[f].reshape(127, 508), frames[f, 1:, 4:]), f
    finally:
        ctx.set_option("dec_strict", 0)
        ctx.set_option("dec_path", "syndrome")


def test_half_frame_workgroups_equal_whole_frame_workgroups(ctx, oracle):
    """enc_units = half (the FFT encoder in workgroups of one column half: two waves) against the default and the oracle: every row count the
    FFT path serves, a frame count that is not a multiple of anything, and the Rx pipe's three ways into the encoder -- frames in memory
    (rx_direct), the fused framing copy from the stream-order buffer with a frame straddling the calls (rx_direct = 0), the pipelined pipe."""
    import torch

    import sdrdaemon_amd as sd

    F = 37
    x = signals.noise(F * 16129, 4242)
    try:
        for R in (1, 5, 16, 17, 31, 32):
            frames = oracle.framer(nb_fec_blocks=R).write(x)
            frames[:, :, 3] = 0
            ctx.set_option("enc_units", "frame")
            a = sd.fec_encode_frames(ctx, frames, R)
            ctx.set_option("enc_units", "half")
            b = sd.fec_encode_frames(ctx, frames, R)
            assert np.array_equal(a, b), R
            for f in (0, 17, F - 1):
                assert np.array_equal(b[f], oracle.frame_encode(frames[f], R)), (R, f)
        S, n = 3, (1 << 22) + 4 * 777
        xs = torch.from_numpy(np.stack([signals.noise(n, 60 + s) for s in range(S)])).cuda()
        cut = (n // 3 + 40) & ~3
        for direct, pipelined in ((1, False), (0, False), (1, True), (0, True)):
            res = []
            for units in ("frame", "half"):
                ctx.set_option("enc_units", units)
                ctx.set_option("rx_direct", direct)
                rx = sd.RxPipe(ctx, S, log2decim=4, nb_fec=32, pipelined=pipelined)
                parts = [rx.process(xs[:, :cut], 1, 2), rx.process(xs[:, cut:], 3, 4)]
                if pipelined:
                    parts.append(torch.from_numpy(rx.flush()).cuda())
                res.append(torch.cat([p for p in parts if p.shape[1]], dim=1).clone())
            ctx.synchronize()
            assert torch.equal(res[0], res[1]), (direct, pipelined)
    finally:
        ctx.set_option("enc_units", "frame")
        ctx.set_option("rx_direct", 1)
