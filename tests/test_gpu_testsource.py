"""GPU TestSource bank: the reference's configuration semantics (TestSource.cpp:59-258) and the
library's integer-exact NCO, bit-exact against its independent statement in oracle/sdr_oracle.c."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sdrdaemon_amd as sd

    assert sd.device_count() > 0
    return sd.Context(0)


def test_generator_matches_the_oracle_and_continues_across_calls(ctx, oracle):
    import sdrdaemon_amd as sd

    ts = sd.TestSource(ctx, 3)
    assert ts.configure("srate=10000000,dfp=100000,power=20", 0)       # README.md:362 style test signal
    assert ts.configure({"srate": 2400000, "dfn": 37000, "power": 0}, 1)  # full scale, negative offset
    assert ts.configure("srate=8000&dfp=4000&power=3", 2)              # '&' separated, offset = srate / 2
    exp_par = [(oracle.lib.orc_nco_phase_inc(100000, 10000000), oracle.lib.orc_nco_amp_q15(20)),
               (oracle.lib.orc_nco_phase_inc(-37000, 2400000), oracle.lib.orc_nco_amp_q15(0)),
               (oracle.lib.orc_nco_phase_inc(4000, 8000), oracle.lib.orc_nco_amp_q15(3))]
    phases = [0, 0, 0]
    for n in (65536, 4099, 1, 70000):  # ragged calls: the phase carries over
        y = ts.read(n)
        ctx.synchronize()
        y = y.cpu().numpy()
        for s in range(3):
            e, phases[s] = oracle.testsource_generate(phases[s], exp_par[s][0], exp_par[s][1], n)
            assert np.array_equal(y[s], e), (n, s, np.argwhere(y[s] != e)[:3])
    # host-memory form of the same call
    yh = ts.read(1000, host=True)
    for s in range(3):
        e, phases[s] = oracle.testsource_generate(phases[s], exp_par[s][0], exp_par[s][1], 1000)
        assert np.array_equal(yh[s], e)


def test_nco_is_a_clean_carrier(ctx, oracle):
    """sanity of the definition itself: within 1.5 LSB of the ideal carrier at the table's 12-bit phase resolution"""
    import sdrdaemon_amd as sd

    ts = sd.TestSource(ctx, 1)
    assert ts.configure("srate=10000000,dfp=100000,power=6")
    n = 100000
    y = ts.read(n, host=True).astype(np.float64)
    inc = oracle.lib.orc_nco_phase_inc(100000, 10000000)
    ph = (np.arange(n, dtype=np.uint64) * inc) & 0xFFFFFFFF
    idx = (ph >> 20).astype(np.float64)
    a = oracle.lib.orc_nco_amp_q15(6)
    assert abs(a / 32768.0 - 10 ** (-6 / 20.0)) < 1e-4
    ideal = np.stack([a * np.cos(2 * np.pi * idx / 4096), a * np.sin(2 * np.pi * idx / 4096)], axis=1)
    assert np.max(np.abs(y - ideal)) <= 1.5
    # and the carrier sits where it should: FFT peak at +100 kHz
    z = y[:65536, 0] + 1j * y[:65536, 1]
    k = int(np.argmax(np.abs(np.fft.fft(z))))
    assert abs(k * 10e6 / 65536 - 100e3) < 10e6 / 65536


def test_configure_mirrors_the_reference(ctx):
    """range checks and error strings of TestSource::configure (TestSource.cpp:71-196), the tuner offset of fcpos
    (:199-209), the block length clamp (:246-251) and the two quirks documented in sdrhip_testsource.cpp"""
    import sdrdaemon_amd as sd

    ts = sd.TestSource(ctx, 2)
    g = ts.get(0)
    assert (g["sample_rate"], g["frequency"], g["block_length"], g["fcpos"]) == (64000, 435000000, 65536, 2)
    for msg, err in (("srate=7999", "Invalid sample rate"), ("srate=10000001", "Invalid sample rate"), ("freq=9999", "Invalid frequency"),
                     ("srate=48000,dfp=24001", "Invalid positive carrier offset"), ("dfn=-1", "Invalid negative carrier offset"),
                     ("power=-1", "Invalid peak power"), ("fcpos=3", "Invalid center frequency position"),
                     ("decim=7", "Invalid log2 decimation factor")):
        assert not ts.configure(msg)
        assert err in ts.error(), msg
    assert ts.get(0)["sample_rate"] == 64000  # a rejected message changes nothing
    assert ts.configure("srate=1000000,freq=144000000,fcpos=0,blklen=100,decim=4", 0)
    g = ts.get(0)
    assert g == {"sample_rate": 1000000, "frequency": 144250000, "block_length": 4096, "decim": 4, "fcpos": 0}  # tuned + srate / 4
    assert ts.configure("fcpos=1,blklen=99999999", 0)
    assert ts.get(0)["frequency"] == 143750000 and ts.get(0)["block_length"] == 1024 * 1024
    assert ts.get(1)["sample_rate"] == 64000  # the other stream is untouched
    assert ts.configure("srate=250000")       # stream = -1: every stream
    assert ts.get(0)["sample_rate"] == ts.get(1)["sample_rate"] == 250000
    # the reference stores m_fcPos before it checks decim (TestSource.cpp:175-197): a valid fcpos survives an invalid decim ...
    assert not ts.configure("fcpos=2,decim=9") and "Invalid log2 decimation factor" in ts.error()
    assert ts.get(0)["fcpos"] == ts.get(1)["fcpos"] == 2 and ts.get(0)["decim"] == 4
    # ... and a message for every stream that one stream must reject (dfp beyond ITS srate / 2) changes none of them
    assert ts.configure("srate=48000", 1)
    assert not ts.configure("dfp=100000,power=3") and "Invalid positive carrier offset" in ts.error()
    assert ts.get(0)["sample_rate"] == 250000 and ts.get(1)["sample_rate"] == 48000


def test_rx_pipe_fed_by_the_bank(ctx, oracle):
    """config 3 without an H2D copy: bank -> decimate16_cen -> frames; the samples the pipe saw are the oracle's"""
    import sdrdaemon_amd as sd

    ts = sd.TestSource(ctx, 2)
    assert ts.configure("srate=10000000,dfp=100000,power=20")
    assert ts.configure("dfn=250000", 1)
    n = 16129 * 16 * 2 + 4096
    x = ts.read(n)
    rx = sd.RxPipe(ctx, 2, log2decim=4, fcpos=sd.FC_CEN, nb_fec=8)
    fr = rx.process(x, 5, 6)
    ctx.synchronize()
    fr = fr.cpu().numpy()
    xs = x.cpu().numpy()
    for s in range(2):
        y, _ = oracle.decimators(0).decimate(4, 2, 16, xs[s])
        f = oracle.framer(nb_fec_blocks=8, tv_sec=5, tv_usec=6)
        e = f.write(y)
        assert fr.shape[1] == e.shape[0] == 2
        for k in range(2):
            assert np.array_equal(fr[s, k, :128], e[k]), (s, k)
