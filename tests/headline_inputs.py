"""Inputs of the benchmarked Tx launch (BASELINE configs[3]) -- shared by bench.py, tests/test_gpu_headline.py and
tests/golden/make_golden.py so that all three mean the same bytes.

The received frames are config 3's output: bench.py's bank (signals.hash_noise streams) through decimate16_cen + UDPSinkFEC
framing + CM256 128+32, the first TX_FRAMES frames of every stream, 24 of each frame's 160 blocks lost (a different random
set per frame: tx_keep_sets), the first 128 survivors in index order."""
import numpy as np

TX_FRAMES = 128        # frames per stream and step
TX_KEEP_SEED = 3       # np.random.RandomState seed of the loss patterns
TX_LOG2_INTERP = 4


# ---- the TestSource-shaped input of the headline step (every BASELINE config names TestSource): 10 Msps, a CW carrier 20 dB under
# full scale at +100 kHz + 1 kHz x (seed mod 1000) (README.md:362's test signal, one offset per stream).  The reference's own
# generator is a float phasor under -ffast-math (not reproducible): the samples are the integer NCO the
# library's TestSource bank defines, restated in oracle/sdr_oracle.c (orc_testsource_generate) -- bench.py gets them from the bank
# on the device, tests/golden/make_golden.py from the oracle on the host, and the digests meet.
TS_SRATE, TS_POWER_DB = 10000000, 20


def ts_offset_hz(seed):
    return 100000 + 1000 * (seed % 1000)


def ts_config_string(seed):
    """the TestSource::configure message (TestSource.cpp:59-215 keys) of the stream with this seed"""
    return "srate=%d,dfp=%d,power=%d" % (TS_SRATE, ts_offset_hz(seed), TS_POWER_DB)


def ts_oracle_samples(orc, n, seed):
    """(n, 2) int16: the stream's first n samples by the oracle's NCO (phase 0 at the first sample)"""
    x, _ = orc.testsource_generate(0, orc.lib.orc_nco_phase_inc(ts_offset_hz(seed), TS_SRATE), orc.lib.orc_nco_amp_q15(TS_POWER_DB), n)
    return x


def tx_keep_sets(nframes_total, seed=TX_KEEP_SEED):
    """(nframes_total, 128) int64: the block indices (ascending) that arrive of every frame: 136 of 160 survive, the collector
    takes the first 128 (SDRdaemonFECBuffer.cpp:143-166)"""
    rs = np.random.RandomState(seed)
    return np.stack([np.sort(rs.permutation(160)[:136])[:128] for _ in range(nframes_total)])


def tx_received_frames(ctx, x, meta, nframes=None, seed=TX_KEEP_SEED):
    """x: (S, n, 2) int16 device tensor, meta: the Rx pipe settings (headline_golden.json "meta") -> (rxf (S, nframes, 128,
    512) uint8 device tensor, keep (S * nframes, 128)).  The Rx side runs on the GPU (its frames are pinned to the reference
    by the Rx digests of the same file)."""
    import torch

    import sdrdaemon_amd as sd

    S = x.shape[0]
    rx = sd.RxPipe(ctx, S, log2decim=4, fcpos=sd.FC_CEN, hb_variant=sd.HB_EO1, sample_bits=16, nb_fec=meta["nb_fec"],
                   center_frequency_khz=meta["center_frequency_khz"], sample_rate=meta["sample_rate"])
    fr = rx.process_view(x, tv_sec=meta["tv_sec"], tv_usec=meta["tv_usec"]).torch()
    if nframes is None:
        nframes = min(TX_FRAMES, fr.shape[1])  # (bench.py --log2-samples below 25: fewer frames per stream)
    assert 0 < nframes <= fr.shape[1], "the step must complete %d frames per stream" % nframes
    keep = tx_keep_sets(S * nframes, seed)
    allb = fr[:, :nframes].reshape(S * nframes, 128 + meta["nb_fec"], 512)
    rxf = allb[torch.arange(S * nframes, device=x.device)[:, None], torch.from_numpy(keep).to(x.device)].contiguous()
    ctx.synchronize()
    del rx
    return rxf.reshape(S, nframes, 128, 512), keep
