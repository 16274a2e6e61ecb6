#!/usr/bin/env python3
"""Randomised parity hunt on the GPU box: random configurations and ragged call sequences through the C ABI against the oracle.
usage: python tools/fuzz_gpu.py [seconds] [seed] [max iterations, 0 = none]
(tests/test_gpu_fuzz_slice.py runs a fixed-seed, iteration-capped slice of it inside `pytest -m gpu`)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import sdrdaemon_amd as sd  # noqa: E402
import signals  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_it = int(sys.argv[3]) if len(sys.argv) > 3 else 0
orc = Oracle()
ctx = sd.Context(0)
t0 = time.time()
stats = {"decim": 0, "interp": 0, "rx": 0, "tx": 0, "fec": 0}
cm = sd.CM256(ctx)


def sizes(rs, scale, k):
    out = []
    for _ in range(k):
        c = rs.randint(0, 6)
        if c == 0:
            out.append(int(rs.randint(0, 5)))
        elif c == 1:
            out.append(int(rs.randint(1, 300)))
        elif c == 2:
            out.append(int(scale * rs.randint(1, 9) + rs.randint(0, 4)))
        else:
            out.append(int(rs.randint(1, 40000) * (1 + 7 * (c == 5))))
        if rs.rand() < 0.06:
            out[-1] = int(rs.randint(300000, 2500000))  # many segments / many matrix-core waves
    return out


it = 0
while time.time() - t0 < budget and (max_it == 0 or it < max_it):
    it += 1
    rs = np.random.RandomState(seed0 * 100003 + it)
    what = rs.choice(["decim", "interp", "rx", "tx", "fec"], p=[0.3, 0.2, 0.2, 0.1, 0.2])
    if os.environ.get("FUZZ_VERBOSE"):  # (a fault kills the process: the last line names the iteration)
        ctx.synchronize()
        print("it", it, what, flush=True)
    # decimator kernel selection per iteration: the VALU cascade, the matrix-core cascade (short spans so that small
    # inputs run many waves + the VALU head / tail pieces), or the library's own choice
    ctx.set_option("decim_path", str(rs.choice(["auto", "valu", "mfma", "mfma"])))
    ctx.set_option("mfma_span", 1024 * int(rs.choice([1, 1, 2, 3, 5, 16])))
    ctx.set_option("rx_fused", int(rs.choice([0, 1, 3])))  # pipelined Rx: the waiting encode in its own launch / inside the decimator's / on the second stream
    ctx.set_option("mfma_ring", int(rs.choice([2, 3, 4, 4])))  # LDS-DMA ring depth of the decimate16 matrix-core kernel (2 = two waves per SIMD)
    ctx.set_option("tx_overlap", int(rs.randint(0, 2)))      # pipelined Tx: decode on the second stream / on the first
    ctx.set_option("rx_direct", int(rs.randint(0, 2)))       # matrix-core decimator inside the Rx pipe: frame-layout stores / stream order + framing pass
    ctx.set_option("enc_path", str(rs.choice(["fft", "fft", "karatsuba"])))  # CM256 128 + R encoder and the syndrome decoder's walk
    ctx.set_option("dec_path", str(rs.choice(["syndrome", "syndrome", "dense"])))
    ctx.set_option("tx_gather", int(rs.choice([1, 1, 0])))                     # Tx pipe on K5w with the fused-plan decoder: no copy of the received originals / payload buffer in between
    ctx.set_option("dec_plan", str(rs.choice(["fused", "fused", "kernel"])))    # dec_max_rows <= 32 on the FFT decoder: the plan inside the decoder's launch / gf_decode_plan_kernel in front
    ctx.set_option("interp_path", str(rs.choice(["wave", "wave", "valu"])))     # K5w (default) / K5
    ctx.set_option("interp_span", int(rs.choice([0, 0, 128, 256, 640, 2048])))  # segment length in inputs (0 = the planner's)
    ctx.set_option("enc_units", str(rs.choice(["frame", "frame", "half"])))       # FFT encoder: a workgroup per frame / per column half of a frame
    if what == "decim":
        S = int(rs.randint(1, 4))
        bias = int(rs.randint(0, 2))
        bits = int(rs.choice([8, 12, 16]))
        d = sd.Decimators(ctx, S, bias)
        ods = [orc.decimators(bias) for _ in range(S)]
        for n in sizes(rs, 2048, int(rs.randint(1, 6))):
            L = int(rs.randint(0, 7))
            fc = int(rs.randint(0, 3))
            x = np.stack([signals.noise(n, int(rs.randint(1 << 30)), bits) if rs.rand() < 0.6 else signals.mixed(n, int(rs.randint(1 << 30))) >> (16 - bits)
                          for _ in range(S)]).astype(np.int16) if n else np.zeros((S, 0, 2), np.int16)
            if n and rs.rand() < 0.4:  # device-memory path: rows padded to a multiple of 4 samples
                import torch

                pad = (n + 3) & ~3
                xt = torch.zeros((S, pad, 2), dtype=torch.int16, device="cuda")
                xt[:, :n] = torch.from_numpy(x).cuda()
                yt, ss = d.decimate(L, fc, bits, xt[:, :n])
                ctx.synchronize()
                y = yt.cpu().numpy()
            else:
                y, ss = d.decimate(L, fc, bits, x)
            y = np.asarray(y).reshape(S, -1, 2)
            for s in range(S):
                e, es = ods[s].decimate(L, fc, bits, x[s])
                assert es == ss and np.array_equal(y[s], e), ("decim", it, L, fc, bias, bits, n, s)
    elif what == "interp":
        S = int(rs.randint(1, 4))
        u = sd.Interpolators(ctx, S)
        ous = [orc.interpolators() for _ in range(S)]
        for n in sizes(rs, 512, int(rs.randint(1, 5))):
            L = int(rs.randint(0, 7))
            n = min(n, 60000)
            x = np.stack([signals.noise(n, int(rs.randint(1 << 30))) for _ in range(S)]) if n else np.zeros((S, 0, 2), np.int16)
            if n and rs.rand() < 0.4:
                import torch

                pad = (n + 3) & ~3
                xt = torch.zeros((S, pad, 2), dtype=torch.int16, device="cuda")
                xt[:, :n] = torch.from_numpy(x).cuda()
                yt = u.interpolate(L, xt[:, :n])
                ctx.synchronize()
                y = yt.cpu().numpy().reshape(S, -1, 2)
            else:
                y = np.asarray(u.interpolate(L, x)).reshape(S, -1, 2)
            for s in range(S):
                assert np.array_equal(y[s], ous[s].interpolate(L, x[s])), ("interp", it, L, n, s)
    elif what == "rx":
        S = int(rs.randint(1, 3))
        L = int(rs.randint(0, 5))
        fc = int(rs.randint(0, 3))  # (decim 0 and inf / sup 1-2 are the filter-less settings framed by K2)
        R = int(rs.choice([0, 1, 7, 13, 32, 100]))
        bias = int(rs.randint(0, 2))
        pipelined = bool(rs.rand() < 0.4)  # frames one call late (the encoder rides in the next call's decimator launch) + flush
        rx = sd.RxPipe(ctx, S, log2decim=L, fcpos=fc, hb_variant=bias, nb_fec=R, pipelined=pipelined)
        waiting = None  # pipelined: what the NEXT call (or the flush) has to deliver: (expected frames per stream, R)
        use_async = bool(rs.rand() < 0.3)  # host calls through sdrhip_rx_submit + sdrhip_rx_collect (one block per batch)
        if use_async:
            rx.set_async(depth=2, blocks=1)

        def check(got, exp, Rexp, tag):
            for s in range(S):
                assert got.shape[1] == exp[s].shape[0], ("rx count",) + tag + (s,)
                for f in range(exp[s].shape[0]):
                    assert np.array_equal(got[s, f, :128], exp[s][f]), ("rx frame",) + tag + (s, f)
                    if Rexp:
                        assert np.array_equal(got[s, f, 128:], orc.frame_encode(exp[s][f], Rexp)), ("rx fec",) + tag + (s, f)

        ods = [orc.decimators(bias) for _ in range(S)]
        frs = [orc.framer(nb_fec_blocks=R) for _ in range(S)]
        dev_rate = 625000 << L  # the pipe is created with the sink rate 625000: the device runs at 625000 * 2^decim
        for k in range(int(rs.randint(1, 6))):
            if k and rs.rand() < 0.3:  # control message between batches
                if pipelined and waiting is not None:  # (the slots change size with fecblk: deliver what waits first)
                    check(rx.flush().reshape(S, -1, 128 + waiting[1], 512), waiting[0], waiting[1], (it, k, "flush"))
                    waiting = None
                L = int(rs.randint(0, 5))
                fc = int(rs.randint(0, 3))
                R = int(rs.choice([0, 1, 7, 13, 32, 100]))
                assert rx.configure({"decim": L, "fcpos": fc, "fecblk": R}), rx.error()
                for fr in frs:
                    fr.s.nb_fec_blocks = R
                    fr.s.sample_rate = dev_rate >> L  # recomputed for every block (sdrdaemonrx.cpp:640-644)
            nd = int(rs.choice([3, 500, 16129, 16130, 8000, 40000, 70000, 150000 >> max(L - 2, 0)]))
            x = np.stack([signals.noise(nd << L, int(rs.randint(1 << 30))) for _ in range(S)])
            if rs.rand() < 0.3:
                import torch

                n_raw = x.shape[1]
                xt = torch.zeros((S, (n_raw + 3) & ~3, 2), dtype=torch.int16, device="cuda")
                xt[:, :n_raw] = torch.from_numpy(x).cuda()
                got = rx.process_view(xt[:, :n_raw], tv_sec=k, tv_usec=it).torch().cpu().numpy().reshape(S, -1, 128 + R, 512)
            elif use_async:
                rx.submit(x, tv_sec=k, tv_usec=it)
                got = rx.collect(wait=True, max_frames=(nd // 16129) + 3).reshape(S, -1, 128 + R, 512)
            else:
                got = rx.process(x, tv_sec=k, tv_usec=it).reshape(S, -1, 128 + R, 512)
            exp = []
            for s in range(S):
                y, ss = ods[s].decimate(L, fc, 16, x[s])
                frs[s].s.sample_bytes, frs[s].s.sample_bits = (ss - 1) // 8 + 1, ss
                frs[s].s.tv_sec, frs[s].s.tv_usec = k, it
                exp.append(frs[s].write(y))
            if not pipelined:
                check(got, exp, R, (it, k))
            else:
                if waiting is not None:
                    check(got, waiting[0], waiting[1], (it, k, "late"))
                else:
                    assert got.shape[1] == 0, ("rx pipelined first", it, k)
                waiting = (exp, R) if exp[0].shape[0] else None
        if pipelined and waiting is not None:
            check(rx.flush().reshape(S, -1, 128 + waiting[1], 512), waiting[0], waiting[1], (it, "end"))
    elif what == "fec":
        # generic CM256 geometry through the single-call ABI: encode, lose blocks, decode in place
        k = int(rs.randint(1, 201))
        m = int(rs.randint(1, min(256 - k, 64) + 1))
        bb = int(rs.choice([1, 3, 4, 16, 127, 508, 509, 700, int(rs.randint(1, 2000))]))
        orig = rs.randint(0, 256, (k, bb)).astype(np.uint8)
        rc, rec = cm.cm256_encode((k, m, bb), orig)
        assert rc == 0 and np.array_equal(rec, orc.cm256_encode(orig, m)), ("fec encode", it, k, m, bb)
        nlost = int(rs.randint(0, min(k, m) + 1))
        lost = sorted(rs.choice(k, nlost, replace=False).tolist())
        recs = sorted(rs.choice(m, nlost, replace=False).tolist())  # recovery rows that replace them, at the lost positions
        data = orig.copy()
        idx = np.arange(k)
        for p_, r_ in zip(lost, recs):
            data[p_] = rec[r_]
            idx[p_] = k + r_
        if rs.rand() < 0.5:  # arrival order is arbitrary
            perm = rs.permutation(k)
            data, idx = np.ascontiguousarray(data[perm]), idx[perm]
        d1, d2 = data.copy(), data.copy()
        rc1, i1 = cm.cm256_decode((k, m, bb), d1, idx)
        rc2, i2 = orc.cm256_decode(d2, idx, k, m)
        assert rc1 == rc2 and np.array_equal(i1, i2) and np.array_equal(d1, d2), ("fec decode", it, k, m, bb, nlost)
        if rc1 == 0 and not (m == 1 and nlost):  # (m == 1: the library's XOR shortcut assumes row k)
            assert np.array_equal(d1[np.argsort(i1)], orig), ("fec roundtrip", it, k, m, bb, nlost)
    else:
        F = int(rs.randint(1, 4)) if rs.rand() < 0.95 else int(rs.randint(70, 160))  # > 64 distinct patterns: plan-cache recycling
        R = int(rs.choice([8, 32, 64]))
        L = int(rs.randint(0, 7))
        x = signals.noise(F * 16129, int(rs.randint(1 << 30)))
        frames = orc.framer(nb_fec_blocks=R).write(x)
        rxb = np.zeros((F, 128, 512), np.uint8)
        # dec_max_rows: the sender's fecblk, the default (128), or a promise some frames BREAK (they must come out as received,
        # holes = 0, and be counted: sdrhip_ctx_get_counter)
        max_rows = int(rs.choice([R, R, 128, max(1, R // 4)]))
        ctx.set_option("dec_max_rows", max_rows)
        exceeded0, n_exceed = ctx.counter("dec_rows_exceeded"), 0
        for f in range(F):
            allb = np.concatenate([frames[f], orc.frame_encode(frames[f], R)])
            nlost = int(rs.randint(0, R + 1))
            lost = set(rs.choice(128 + R, nlost, replace=False).tolist())
            if sum(1 for i in lost if i < 128) == 1:
                lost.discard(128)  # cm256's RecoveryCount == 1 shortcut only works with recovery row 128 (mirrored quirk, tested elsewhere)
            keep = [i for i in range(128 + R) if i not in lost][:128]
            if rs.rand() < 0.12:
                # a frame that never reaches 128 blocks (the reference emits it with holes, SDRdaemonFECBuffer.cpp:72-75,
                # 109): the batched API wants 128 entries, the caller pads with a repeat -> not decodable -> zeros in the holes
                keep = [i for i in keep if i < 128][:int(rs.randint(1, 120))]
                hole = np.zeros((127, 508), np.uint8)
                for b in keep:
                    if b >= 1:
                        hole[b - 1] = allb[b, 4:]
                x[f * 16129:(f + 1) * 16129] = hole.reshape(-1).view(np.int16).reshape(-1, 2)
                keep = keep + [keep[0]] * (128 - len(keep))
            elif sum(1 for i in keep if i >= 128) > max_rows:
                n_exceed += 1
                hole = np.zeros((127, 508), np.uint8)
                for b in keep:
                    if 1 <= b < 128:
                        hole[b - 1] = allb[b, 4:]
                x[f * 16129:(f + 1) * 16129] = hole.reshape(-1).view(np.int16).reshape(-1, 2)
            rxb[f] = allb[keep]
        tx_pipelined = bool(rs.rand() < 0.4)  # samples one call late (decode beside the previous batch's interpolator) + flush
        tx = sd.TxPipe(ctx, 1, L, pipelined=tx_pipelined)
        cut = int(rs.randint(1, F)) if (tx_pipelined and F > 1) else F  # pipelined: two batches, the second one delivers the first
        if rs.rand() < 0.4:  # frames resident on the device: the planning kernel reads the block indices from the headers
            import torch

            rxt = torch.from_numpy(rxb).cuda()
            parts = [tx.process(rxt[:cut])] + ([tx.process(rxt[cut:])] if cut < F else [])
            if tx_pipelined:
                parts.append(tx.flush(device=rxt.device)[0])
            ctx.synchronize()
            y = np.concatenate([p_.cpu().numpy().reshape(-1, 2) for p_ in parts])
        else:
            parts = [tx.process(rxb[:cut])] + ([tx.process(rxb[cut:])] if cut < F else [])
            if tx_pipelined:
                parts.append(tx.flush()[0])
            y = np.concatenate([np.asarray(p_).reshape(-1, 2) for p_ in parts])
        assert np.array_equal(y, orc.interpolators().interpolate(L, x)), ("tx", it, F, R, L, max_rows)
        assert ctx.counter("dec_rows_exceeded") - exceeded0 == n_exceed, ("tx dec_max_rows counter", it, max_rows, n_exceed)
        ctx.set_option("dec_max_rows", 128)
    stats[what] += 1
print("fuzz OK: %d iterations in %.0f s: %s" % (it, time.time() - t0, stats))
