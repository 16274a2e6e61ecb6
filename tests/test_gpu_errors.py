"""Error behaviour of the C ABI on a live GPU: the argument checks of Downsampler::configure /
Upsampler::configure (Downsampler.cpp:39-59, Upsampler.cpp:38-42), alignment, strides, empty calls."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sdrdaemon_amd as sd

    assert sd.device_count() > 0
    return sd.Context(0)


def test_invalid_arguments_are_rejected_with_messages(ctx):
    import sdrdaemon_amd as sd

    d = sd.Decimators(ctx, 1, 0)
    x = np.zeros((64, 2), np.int16)
    for log2, fc, msg in ((7, 2, "Invalid log2 decimation factor"), (-1, 2, "Invalid log2 decimation factor"),
                          (4, 3, "Invalid Fc position index")):
        with pytest.raises(sd.SdrHipError) as e:
            d.decimate(log2, fc, 16, x)
        assert e.value.code == -1 and msg in str(e.value)
    with pytest.raises(sd.SdrHipError):
        d.decimate(4, 2, 17, x)
    u = sd.Interpolators(ctx, 1)
    with pytest.raises(sd.SdrHipError) as e:
        u.interpolate(7, x)
    assert "Invalid log2 interpolation factor" in str(e.value)
    with pytest.raises(sd.SdrHipError):
        sd.Decimators(ctx, 1, 5)
    sd.RxPipe(ctx, 1, log2decim=0)  # (the filter-less settings are part of the pipe since K2 frames them)
    with pytest.raises(sd.SdrHipError):
        sd.RxPipe(ctx, 1, log2decim=7)
    with pytest.raises(sd.SdrHipError):
        sd.RxPipe(ctx, 1, nb_fec=129)


def test_downsampler_upsampler_configure_mirror_the_reference(ctx):
    import sdrdaemon_amd as sd

    dn = sd.Downsampler(ctx)
    assert dn.configure({"decim": "4", "fcpos": "2"}) and dn.getLog2Decimation() == 4
    assert not dn.configure({"decim": "7"}) and dn.error() == "Invalid log2 decimation factor"
    assert not dn.configure({"fcpos": "3"}) and dn.error() == "Invalid Fc position index"
    up = sd.Upsampler(ctx)
    assert up.configure({"interp": "6"}) and up.getLog2Interpolation() == 6
    assert not up.configure({"interp": "9"}) and up.error() == "Invalid log2 interpolation factor"
    x = np.arange(256, dtype=np.int16).reshape(128, 2)
    y, ss = dn.process(16, x)  # decim 4, centred
    assert y.shape == (8, 2) and ss == 16
    assert np.array_equal(sd.Upsampler(ctx, 0).process(x), x)  # interp 0 copies (Upsampler.cpp:54-57)
    r, ss = dn.rescale(12, x)  # decimate1: << 4, sampleSize unchanged (Decimators.cpp:22-35)
    assert ss == 12 and np.array_equal(r, (x.astype(np.int32) << 4).astype(np.int16))


def test_misaligned_device_pointer_is_an_error_not_a_crash(ctx):
    import torch

    import sdrdaemon_amd as sd

    lib = ctx.lib
    d = sd.Decimators(ctx, 1, 0)
    x = torch.zeros((1024 + 1, 2), dtype=torch.int16, device="cuda")
    out = torch.zeros((64, 2), dtype=torch.int16, device="cuda")
    ss, n_out = C.c_uint(16), C.c_size_t(0)
    rc = lib.sdrhip_decimate(d.h, 4, 2, C.byref(ss), C.c_void_p(x.data_ptr() + 4), 1024, 1024, C.c_void_p(out.data_ptr()), 64,
                             C.byref(n_out), sd.MEM_DEVICE)
    assert rc == -4 and b"16-byte aligned" in lib.sdrhip_last_error()


def test_empty_and_sub_block_calls(ctx):
    import sdrdaemon_amd as sd

    d = sd.Decimators(ctx, 1, 0)
    y, ss = d.decimate(4, 2, 16, np.zeros((0, 2), np.int16))
    assert y.shape == (0, 2) and ss == 16
    y, ss = d.decimate(4, 2, 16, np.ones((15, 2), np.int16))  # < 16 samples: nothing consumed
    assert y.shape == (0, 2)
    u = sd.Interpolators(ctx, 1)
    assert u.interpolate(4, np.zeros((0, 2), np.int16)).shape == (0, 2)
    rx = sd.RxPipe(ctx, 1)
    assert rx.process(np.zeros((160, 2), np.int16)).shape[0] == 0


def test_context_outlives_its_handles_in_any_destruction_order():
    """sdrhip_ctx_destroy before sdrhip_decimators_destroy (what a garbage collector may do) must not
    crash: handles keep their context alive."""
    import sdrdaemon_amd as sd

    c2 = sd.Context(0)
    d = sd.Decimators(c2, 2, 0)
    u = sd.Interpolators(c2, 1)
    rx = sd.RxPipe(c2, 1)
    lib = c2.lib
    lib.sdrhip_ctx_destroy(c2.h)  # context first
    c2.h = C.c_void_p()
    y, _ = d.decimate(4, 2, 16, np.ones((2, 64, 2), np.int16))  # still usable
    assert y.shape == (2, 4, 2)
    for obj in (rx, u, d):
        obj.close()


def test_device_call_too_short_for_any_output_needs_no_output_buffer(ctx):
    """3 samples through decimate16 on device memory: torch hands out a NULL pointer for the empty result."""
    import torch

    import sdrdaemon_amd as sd

    d = sd.Decimators(ctx, 2, 0)
    x = torch.zeros((2, 4, 2), dtype=torch.int16, device="cuda")
    y, ss = d.decimate(4, 2, 16, x[:, :3])
    assert tuple(y.shape) == (2, 0, 2) and ss == 16


def test_create_use_destroy_cycles_do_not_leak():
    """A daemon reconfigures for months: 120 cycles of context + every handle kind, each used once (device buffers, pinned staging, streams,
    events, the asynchronous rings), then closed.  Device memory and the process's resident set must come back."""
    import gc
    import resource

    import torch

    import sdrdaemon_amd as sd

    x = np.zeros((2, 70000, 2), np.int16)
    x[:, :, 0] = np.arange(70000) % 251

    def cycle(i):
        c = sd.Context(0)
        d = sd.Decimators(c, 2, i & 1)
        y, _ = d.decimate(4, 2, 16, x)
        u = sd.Interpolators(c, 2)
        z = u.interpolate(2, y)
        rx = sd.RxPipe(c, 2, log2decim=1 + i % 4, nb_fec=8 * (i % 5), pipelined=bool(i & 2))
        fr = rx.process(x, 1, 2)
        tx = sd.TxPipe(c, 2, 1 + i % 3)
        if fr.shape[1]:
            tx.process(np.ascontiguousarray(fr[:, :, :128]))
        ts = sd.TestSource(c, 2)
        ts.read(4096)
        cm = sd.CM256(c)
        assert cm.isInitialized()
        c.synchronize()
        for obj in (ts, tx, rx, u, d):
            obj.close()
        c.close()
        return z.shape

    for i in range(8):  # allocator pools, code objects, hipBLAS-free first-use costs
        cycle(i)
    gc.collect()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for i in range(120):
        cycle(i)
    gc.collect()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert free0 - free1 < 64 << 20, "device memory leaked: %.1f MiB over 120 cycles" % ((free0 - free1) / 2**20)
    assert rss1 - rss0 < 96 << 10, "host memory grew by %.1f MiB over 120 cycles (ru_maxrss)" % ((rss1 - rss0) / 1024)
