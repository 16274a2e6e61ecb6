"""Host-side mirror of the reference's interface for the hot path, on top of libsdrhip.so.

Class and method names follow the reference (Decimators.h:32-71, Interpolators.h:35-61,
Downsampler.h:26-83, Upsampler.h:27-70, the CM256 call sites UDPSinkFEC.cpp:195-246 /
SDRdaemonFECBuffer.cpp:148-213) so that the parity tests read like code written against the
reference.  Every method accepts

  * numpy int16 arrays  -> SDRHIP_MEM_HOST (staged through the GPU, synchronous), or
  * torch int16 CUDA tensors -> SDRHIP_MEM_DEVICE (zero-copy, enqueued on the context stream).

Shapes: one stream (n, 2); a bank of S streams (S, n, 2).  There is no CPU implementation
behind these classes: without libsdrhip.so or without a GPU they raise.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import (BLOCK_BYTES, FC_CEN, FC_INF, FC_SUP, HB_DB, HB_EO1, MEM_DEVICE, MEM_HOST, NB_ORIGINAL,  # noqa: F401
                   SAMPLES_PER_FRAME, UDPSIZE, CM256Block, CM256Params, RxConfig, SdrHipError, check)

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


K_DECIMATE, K_INTERPOLATE, K_FEC_ENCODE, K_FEC_DECODE = 0, 1, 2, 3


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def device_count():
    return _lib.lib().sdrhip_device_count()


class Context:
    """One per GPU (sdrhip_ctx).  stream: a torch.cuda.Stream, a raw hipStream_t int, or None
    (torch's current stream when torch sees the device, else the null stream)."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.lib()
        self.device = device
        if stream is None and torch is not None and torch.cuda.is_available():
            with torch.cuda.device(device):
                stream = torch.cuda.current_stream().cuda_stream
        elif stream is not None and hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        self.h = C.c_void_p()
        self.options = {}
        check(self.lib.sdrhip_ctx_create(device, C.c_void_p(stream or 0), C.byref(self.h)))

    def synchronize(self):
        check(self.lib.sdrhip_ctx_synchronize(self.h))

    def set_option(self, key, value):
        """kernel-path knobs for tests and tools (sdrhip_ctx_set_option): decim_path, mfma_span, mfma_min, interp_path,
        interp_span, rx_fused; the defaults were read from the SDRHIP_* environment when the context was created"""
        check(self.lib.sdrhip_ctx_set_option(self.h, str(key).encode(), str(value).encode()))
        self.options[str(key)] = str(value)

    def option(self, key, default=None):
        """what this Context was last TOLD for `key`: by set_option, else by the SDRHIP_<KEY> environment variable that the library
        read when the context was created, else `default` (the library's own default is not queried)"""
        if key in self.options:
            return self.options[key]
        return os.environ.get("SDRHIP_" + str(key).upper(), default)

    def host_alloc(self, shape, dtype=np.int16):
        """numpy array on pinned host memory of the library (sdrhip_host_alloc): blocks submitted from it are uploaded in
        place.  Keep the Context alive while the array is in use; free with host_free(array)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.sdrhip_host_alloc(self.h, n)
        if not p:
            raise MemoryError("sdrhip_host_alloc(%d) failed" % n)
        buf = (C.c_char * n).from_address(p)
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._host_allocs = getattr(self, "_host_allocs", {})
        self._host_allocs[a.ctypes.data] = p
        return a

    def host_free(self, a):
        p = getattr(self, "_host_allocs", {}).pop(a.ctypes.data, None)
        if p:
            self.lib.sdrhip_host_free(self.h, C.c_void_p(p))

    def counter(self, key):
        """device-side event counters (sdrhip_ctx_get_counter; synchronises): "dec_rows_exceeded" = frames the batched decoder
        left unrepaired because they carried more recovery blocks than the dec_max_rows option promises"""
        v = C.c_uint64(0)
        check(self.lib.sdrhip_ctx_get_counter(self.h, str(key).encode(), C.byref(v)))
        return v.value

    def timing_begin(self):
        check(self.lib.sdrhip_ctx_timing_begin(self.h))

    def timing_end(self):
        ms = C.c_float(0)
        check(self.lib.sdrhip_ctx_timing_end(self.h, C.byref(ms)))
        return ms.value

    def kernel_timing(self, enable=True):
        check(self.lib.sdrhip_ctx_kernel_timing(self.h, 1 if enable else 0))

    def kernel_timing_read(self, kernel_class):
        """-> (total_ms, launches) of the kernel class since the last read (K_* constants)"""
        ms, n = C.c_double(0), C.c_uint(0)
        check(self.lib.sdrhip_ctx_kernel_timing_read(self.h, kernel_class, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self.h:
            self.lib.sdrhip_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _bank_view(iq, nstreams):
    """-> (array (S, n, 2) contiguous-per-stream, is_torch, squeeze)"""
    squeeze = False
    if _is_torch(iq):
        if iq.dtype != torch.int16 or not iq.is_cuda:
            raise TypeError("torch input must be an int16 CUDA tensor")
        if iq.dim() == 2:
            iq, squeeze = iq.unsqueeze(0), True
        if iq.dim() != 3 or iq.shape[2] != 2 or iq.shape[0] != nstreams:
            raise ValueError("expected shape (%d, n, 2)" % nstreams)
        if iq.stride(2) != 1 or iq.stride(1) != 2:
            iq = iq.contiguous()
        return iq, True, squeeze
    a = np.asarray(iq)
    if a.dtype != np.int16:
        raise TypeError("numpy input must be int16")
    if a.ndim == 2:
        a, squeeze = a[None], True
    if a.ndim != 3 or a.shape[2] != 2 or a.shape[0] != nstreams:
        raise ValueError("expected shape (%d, n, 2)" % nstreams)
    return np.ascontiguousarray(a), False, squeeze


def _ptr(x):
    return C.c_void_p(x.data_ptr()) if _is_torch(x) else C.c_void_p(x.ctypes.data)


def _stride_samples(x):
    return (x.stride(0) // 2) if _is_torch(x) else (x.strides[0] // 4)


def _alloc_like(x, shape, dtype_np=np.int16):
    if _is_torch(x):
        tdt = {np.int16: torch.int16, np.uint8: torch.uint8}[dtype_np]
        # rows padded to 16 bytes so that every stream starts aligned
        return torch.empty(shape, dtype=tdt, device=x.device)
    return np.empty(shape, dtype=dtype_np)


def _plan_dict(fn, h):
    p = _lib.DecimPlan()
    check(fn(h, C.byref(p)))
    return {"path": {0: None, 1: "valu", 2: "mfma"}[p.path], "span": p.span, "wps": p.wps, "npieces": p.npieces, "nseg": p.nseg,
            "head": p.head, "tail_start": p.tail_start}


class Decimators:
    """Bank of reference `Decimators` objects (Decimators.h:32-71)."""

    def __init__(self, ctx, nstreams=1, hb_variant=HB_EO1):
        self.ctx, self.nstreams = ctx, nstreams
        self.h = C.c_void_p()
        check(ctx.lib.sdrhip_decimators_create(ctx.h, nstreams, hb_variant, C.byref(self.h)))

    def reset(self):
        check(self.ctx.lib.sdrhip_decimators_reset(self.h))

    def decimate(self, log2decim, fcpos, sample_size, iq, out=None):
        """Decimators::decimate<2^log2decim>_{inf,sup,cen}(sampleSize, in, out).
        Returns (out, new_sample_size)."""
        x, is_t, squeeze = _bank_view(iq, self.nstreams)
        S, n = x.shape[0], x.shape[1]
        n_res = (n >> log2decim) if 0 <= log2decim <= 6 else 0  # out-of-range factors are rejected by the library
        if out is None:
            if is_t:  # per-stream rows padded to a multiple of 4 samples (16-byte aligned rows)
                pad = (n_res + 3) & ~3
                buf = torch.empty((S, max(pad, 4), 2), dtype=torch.int16, device=x.device)
                out = buf[:, :n_res]
            else:
                out = np.empty((S, n_res, 2), dtype=np.int16)
        if is_t and S > 1 and (x.stride(0) // 2) % 4:
            raise ValueError("device bank input: per-stream stride must be a multiple of 4 samples")
        ss = C.c_uint(sample_size)
        n_out = C.c_size_t(0)
        check(self.ctx.lib.sdrhip_decimate(self.h, log2decim, fcpos, C.byref(ss), _ptr(x), n, _stride_samples(x), _ptr(out),
                                           _stride_samples(out) if S > 1 else n_res, C.byref(n_out),
                                           MEM_DEVICE if is_t else MEM_HOST))
        return (out[0] if squeeze else out), ss.value

    def last_plan(self):
        """what the last cascade launch was (sdrhip_decimators_last_plan): dict with path ('valu' / 'mfma' / None), span, wps, ..."""
        return _plan_dict(self.ctx.lib.sdrhip_decimators_last_plan, self.h)

    def close(self):
        if self.h:
            self.ctx.lib.sdrhip_decimators_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Downsampler:
    """Reference `Downsampler` (Downsampler.h:26-83): configuration + dispatch."""

    def __init__(self, ctx, decim=0, fcpos=FC_CEN, nstreams=1, hb_variant=HB_EO1):
        self.m_decim, self.m_fcPos = decim, fcpos
        self.m_error = ""
        self.m_decimators = Decimators(ctx, nstreams, hb_variant)

    def configure(self, m):
        """m: dict of the parsekv pairs (Downsampler.cpp:32-67).  Returns False and sets error()
        on an invalid value, like the reference."""
        if "decim" in m:
            v = int(m["decim"])
            if v < 0 or v > 6:
                self.m_error = "Invalid log2 decimation factor"
                return False
            self.m_decim = v
        if "fcpos" in m:
            v = int(m["fcpos"])
            if v < FC_INF or v > FC_CEN:
                self.m_error = "Invalid Fc position index"
                return False
            self.m_fcPos = v
        return True

    def getLog2Decimation(self):
        return self.m_decim

    def error(self):
        e, self.m_error = self.m_error, ""
        return e

    def __bool__(self):
        return not self.m_error

    def process(self, sample_size, samples_in):
        """Downsampler::process (Downsampler.cpp:74-162) -> (samples_out, sampleSize)."""
        return self.m_decimators.decimate(self.m_decim, self.m_fcPos, sample_size, samples_in)

    def rescale(self, sample_size, samples_inout):
        """Downsampler::rescale = Decimators::decimate1 (Downsampler.cpp:69-72)."""
        return self.m_decimators.decimate(0, self.m_fcPos, sample_size, samples_inout)


class TestSource:
    """Bank of the reference's TestSource devices (TestSource.h:29-116) generating on the GPU: configure() takes the
    reference's control string / key-value map (TestSource.cpp:59-215), read() returns the next samples of every
    stream as a CUDA tensor (or numpy with host=True).  The sample arithmetic is the library's integer NCO."""

    def __init__(self, ctx, nstreams=1):
        self.ctx, self.nstreams = ctx, nstreams
        self.h = C.c_void_p()
        self.m_error = ""
        check(ctx.lib.sdrhip_testsource_create(ctx.h, nstreams, C.byref(self.h)))

    def configure(self, m, stream=-1):
        """-> bool like TestSource::configure; the message is kept for error()"""
        kv = m if isinstance(m, str) else ",".join("%s=%s" % (k, v) for k, v in m.items())
        try:
            check(self.ctx.lib.sdrhip_testsource_configure(self.h, stream, kv.encode()))
        except SdrHipError as e:
            self.m_error = str(e)
            return False
        return True

    def error(self):
        e, self.m_error = self.m_error, ""
        return e

    def get(self, stream=0):
        sr, fr, bl, dc, fc = C.c_uint32(0), C.c_uint32(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(self.ctx.lib.sdrhip_testsource_get(self.h, stream, C.byref(sr), C.byref(fr), C.byref(bl), C.byref(dc), C.byref(fc)))
        return {"sample_rate": sr.value, "frequency": fr.value, "block_length": bl.value, "decim": dc.value, "fcpos": fc.value}

    def get_sample_rate(self, stream=0):
        return self.get(stream)["sample_rate"]

    def get_frequency(self, stream=0):
        return self.get(stream)["frequency"]

    def read(self, n, host=False, out=None):
        """-> (S, n, 2) int16 (squeezed for one stream)"""
        S = self.nstreams
        pad = (n + 3) & ~3
        if out is None:
            out = np.empty((S, pad, 2), np.int16) if host else torch.empty((S, pad, 2), dtype=torch.int16, device=torch.device("cuda", self.ctx.device))
        check(self.ctx.lib.sdrhip_testsource_read(self.h, _ptr(out), n, out.shape[1], MEM_HOST if host else MEM_DEVICE))
        y = out[:, :n]
        return y[0] if S == 1 else y

    def close(self):
        if self.h:
            self.ctx.lib.sdrhip_testsource_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Interpolators:
    """Bank of reference `Interpolators` objects (Interpolators.h:35-61)."""

    def __init__(self, ctx, nstreams=1):
        self.ctx, self.nstreams = ctx, nstreams
        self.h = C.c_void_p()
        check(ctx.lib.sdrhip_interpolators_create(ctx.h, nstreams, C.byref(self.h)))

    def reset(self):
        check(self.ctx.lib.sdrhip_interpolators_reset(self.h))

    def interpolate(self, log2interp, iq, out=None):
        """Interpolators::interpolate<2^log2interp>_cen(in, out)."""
        x, is_t, squeeze = _bank_view(iq, self.nstreams)
        S, n = x.shape[0], x.shape[1]
        n_res = (n << log2interp) if 0 <= log2interp <= 6 else 0
        if out is None:
            if is_t:
                pad = (n_res + 3) & ~3
                out = torch.empty((S, max(pad, 4), 2), dtype=torch.int16, device=x.device)[:, :n_res]
            else:
                out = np.empty((S, n_res, 2), dtype=np.int16)
        n_out = C.c_size_t(0)
        check(self.ctx.lib.sdrhip_interpolate(self.h, log2interp, _ptr(x), n, _stride_samples(x), _ptr(out),
                                              _stride_samples(out) if S > 1 else n_res, C.byref(n_out),
                                              MEM_DEVICE if is_t else MEM_HOST))
        return out[0] if squeeze else out

    def close(self):
        if self.h:
            self.ctx.lib.sdrhip_interpolators_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Upsampler:
    """Reference `Upsampler` (Upsampler.h:27-70)."""

    def __init__(self, ctx, interp=0, nstreams=1):
        self.m_interp = interp
        self.m_error = ""
        self.m_interpolators = Interpolators(ctx, nstreams)

    def configure(self, m):
        if "interp" in m:
            v = int(m["interp"])
            if v < 0 or v > 6:
                self.m_error = "Invalid log2 interpolation factor"
                return False
            self.m_interp = v
        return True

    def getLog2Interpolation(self):
        return self.m_interp

    def error(self):
        e, self.m_error = self.m_error, ""
        return e

    def process(self, samples_in):
        return self.m_interpolators.interpolate(self.m_interp, samples_in)


class CM256:
    """The `CM256` object of the reference's call sites (UDPSinkFEC.h:123, SDRdaemonFECBuffer.h:173)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def isInitialized(self):
        return True

    def cm256_encode(self, params, originals, recovery_out=None):
        """originals: (k, BlockBytes) uint8 numpy (taken positionally).  Returns (rc, recovery)."""
        k, m, bb = params
        originals = np.ascontiguousarray(originals, dtype=np.uint8)
        blocks = (CM256Block * k)()
        for i in range(k):
            blocks[i].Block = originals[i].ctypes.data
            blocks[i].Index = i
        rec = np.zeros((m, bb), dtype=np.uint8) if recovery_out is None else recovery_out
        rc = self.ctx.lib.sdrhip_cm256_encode(self.ctx.h, CM256Params(k, m, bb), blocks, C.c_void_p(rec.ctypes.data))
        return rc, rec

    def cm256_decode(self, params, data, indices):
        """data: (k, BlockBytes) uint8 received blocks, modified in place; indices: their Index
        fields.  Returns (rc, indices_after) -- the library's in-place contract."""
        k, m, bb = params
        assert data.dtype == np.uint8 and data.flags.c_contiguous and data.shape == (k, bb)
        blocks = (CM256Block * k)()
        for i in range(k):
            blocks[i].Block = data[i].ctypes.data
            blocks[i].Index = int(indices[i])
        rc = self.ctx.lib.sdrhip_cm256_decode(self.ctx.h, CM256Params(k, m, bb), blocks)
        return rc, np.array([blocks[i].Index for i in range(k)], dtype=np.uint8)


def fec_encode_frames(ctx, frames, nb_fec):
    """frames (F, 128, 512) uint8 (numpy or CUDA tensor) -> recovery super blocks (F, nb_fec, 512)."""
    is_t = _is_torch(frames)
    F = frames.shape[0]
    if not is_t:
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
    out = torch.zeros((F, nb_fec, 512), dtype=torch.uint8, device=frames.device) if is_t else np.zeros((F, nb_fec, 512), np.uint8)
    check(ctx.lib.sdrhip_fec_encode_frames(ctx.h, _ptr(frames), F, nb_fec, _ptr(out), MEM_DEVICE if is_t else MEM_HOST))
    return out


def fec_decode_frames(ctx, rx, indices=None, want_block0=False):
    """rx (F, 128, 512) uint8: first 128 received super blocks per frame, arrival order.
    -> payload (F, 127*508) uint8 [, block0 (F, 508)]"""
    is_t = _is_torch(rx)
    F = rx.shape[0]
    if not is_t:
        rx = np.ascontiguousarray(rx, dtype=np.uint8)
    if indices is not None:  # optional: by default the library reads header.blockIndex of the super blocks itself
        indices = np.ascontiguousarray(indices, dtype=np.uint8)
    if is_t:
        payload = torch.empty((F, 127 * 508), dtype=torch.uint8, device=rx.device)
        b0 = torch.empty((F, 508), dtype=torch.uint8, device=rx.device) if want_block0 else None
    else:
        payload = np.empty((F, 127 * 508), np.uint8)
        b0 = np.empty((F, 508), np.uint8) if want_block0 else None
    check(ctx.lib.sdrhip_fec_decode_frames(ctx.h, _ptr(rx), C.c_void_p(indices.ctypes.data if indices is not None else 0), F, _ptr(payload),
                                           _ptr(b0) if want_block0 else C.c_void_p(0), MEM_DEVICE if is_t else MEM_HOST))
    return (payload, b0) if want_block0 else payload


class _DeviceView:
    """A strided uint8 view of library-owned device memory (__cuda_array_interface__ v2)."""

    def __init__(self, ptr, shape, strides, device, owner=None):
        self.shape, self.device = tuple(shape), device
        self._owner = owner  # the handle whose memory this is: stays alive as long as the view (or a tensor made from it) does
        self.__cuda_array_interface__ = {"shape": tuple(shape), "strides": tuple(strides), "typestr": "|u1",
                                         "data": (ptr, False), "version": 2}

    def torch(self):
        """materialise as a torch tensor sharing the memory"""
        if self.shape[1] == 0:
            return torch.empty(self.shape, dtype=torch.uint8, device=self.device)
        return torch.as_tensor(self, device=self.device)


class RxPipe:
    """Downsampler -> UDPSinkFEC framing -> CM256 encode for a bank of streams (sdrhip_rx)."""

    def __init__(self, ctx, nstreams=1, log2decim=4, fcpos=FC_CEN, hb_variant=HB_EO1, sample_bits=16, nb_fec=32,
                 center_frequency_khz=435000, sample_rate=625000, pipelined=False):
        """pipelined: a process() call returns the frames the PREVIOUS call completed (their recovery blocks are computed inside
        this call's decimator launch, sdrhip_rx_set_pipelined); flush() / flush_view() return the last call's at the end."""
        self.ctx, self.nstreams, self.nb_fec = ctx, nstreams, nb_fec
        self.cfg = RxConfig(log2decim, fcpos, hb_variant, sample_bits, nb_fec, center_frequency_khz, sample_rate)
        self.h = C.c_void_p()
        self.m_error = ""
        self.m_device_rate = sample_rate << log2decim  # DeviceSource::get_sample_rate(): the sink gets it >> decim
        check(ctx.lib.sdrhip_rx_create(ctx.h, nstreams, C.byref(self.cfg), C.byref(self.h)))
        self.pipelined = bool(pipelined)
        if pipelined:
            check(ctx.lib.sdrhip_rx_set_pipelined(self.h, 1))

    def error(self):
        e, self.m_error = self.m_error, ""
        return e

    def _view(self, device):
        base, stride, cnt = C.c_void_p(0), C.c_size_t(0), C.c_size_t(0)
        check(self.ctx.lib.sdrhip_rx_frames_view(self.h, C.byref(base), C.byref(stride), C.byref(cnt)))
        fb = (NB_ORIGINAL + self.nb_fec) * UDPSIZE
        return _DeviceView(base.value or 0, (self.nstreams, cnt.value, NB_ORIGINAL + self.nb_fec, UDPSIZE), (stride.value, fb, UDPSIZE, 1), device, owner=self)

    def flush_view(self, device="cuda"):
        """pipelined mode: encode and show (zero copy) the frames the last process() call completed"""
        nf = C.c_size_t(0)
        check(self.ctx.lib.sdrhip_rx_flush(self.h, C.c_void_p(0), 0, C.byref(nf), MEM_DEVICE))
        return self._view(torch.device(device))

    def flush(self):
        """pipelined mode: -> the frames the last process() call completed, as a host array (S, n, 128 + nb_fec, 512)"""
        cap = max(self.max_frames(0), 1)
        fb = (NB_ORIGINAL + self.nb_fec) * UDPSIZE
        out = np.empty((self.nstreams, cap, NB_ORIGINAL + self.nb_fec, UDPSIZE), np.uint8)
        nf = C.c_size_t(0)
        check(self.ctx.lib.sdrhip_rx_flush(self.h, _ptr(out), cap * fb, C.byref(nf), MEM_HOST))
        return out[:, :nf.value]

    def last_plan(self):
        """the decimator launch of the last process() call (sdrhip_rx_last_plan)"""
        return _plan_dict(self.ctx.lib.sdrhip_rx_last_plan, self.h)

    def reconfigure(self, **kw):
        """Live change between two process() calls (sdrhip_rx_reconfigure): any of log2decim, fcpos,
        sample_bits, nb_fec, center_frequency_khz, sample_rate."""
        cfg = RxConfig(self.cfg.log2decim, self.cfg.fcpos, self.cfg.hb_variant, self.cfg.sample_bits, self.cfg.nb_fec,
                       self.cfg.center_frequency_khz, self.cfg.sample_rate)
        for k, v in kw.items():
            if k not in ("log2decim", "fcpos", "sample_bits", "nb_fec", "center_frequency_khz", "sample_rate"):
                raise TypeError("unknown rx setting %r" % k)
            setattr(cfg, k, int(v))
        check(self.ctx.lib.sdrhip_rx_reconfigure(self.h, C.byref(cfg)))
        self.cfg, self.nb_fec = cfg, cfg.nb_fec

    def configure(self, m):
        """The control-message keys of sdrdaemonrx (parsekv pairs): decim, fcpos (Downsampler.cpp:32-67),
        fecblk (UDPSink::setNbBlocksFEC), freq in Hz (setCenterFrequency: kHz on the wire), srate
        (sample rate of the device: the sink gets srate >> decim, sdrdaemonrx.cpp:622-631).  Returns False
        on an invalid value, like Downsampler::configure; unknown keys belong to other components."""
        kw = {}
        try:
            if "decim" in m:
                kw["log2decim"] = int(m["decim"])
            if "fcpos" in m:
                kw["fcpos"] = int(m["fcpos"])
            if "fecblk" in m:
                kw["nb_fec"] = int(m["fecblk"])
            if "freq" in m:
                kw["center_frequency_khz"] = int(m["freq"]) // 1000
            if "srate" in m:
                self.m_device_rate = int(m["srate"])
            if "srate" in m or "decim" in m:
                # the reference recomputes get_sample_rate() / (1 << decim) for every block (sdrdaemonrx.cpp:640-644)
                kw["sample_rate"] = self.m_device_rate >> kw.get("log2decim", self.cfg.log2decim)
            if kw:
                self.reconfigure(**kw)
        except (ValueError, SdrHipError) as e:
            self.m_error = str(e)
            return False
        return True

    def max_frames(self, n_in):
        return self.ctx.lib.sdrhip_rx_max_frames(self.h, n_in)

    # ---- asynchronous host-pointer entry (sdrhip_rx_submit / sdrhip_rx_collect)
    def set_async(self, depth=4, blocks=1):
        """ring of `depth` batches, `blocks` submitted blocks per upload + launch + download"""
        check(self.ctx.lib.sdrhip_rx_set_async(self.h, depth, blocks))
        # samples per stream of the batches not collected yet (oldest first), of the batch being filled and of the batch collected
        # last (a pipelined pipe delivers the PREVIOUS batch's frames): collect() sizes its buffer from them, not from a lifetime total
        self._async_blocks = blocks
        self._async_batches = []
        self._async_fill = [0, 0]
        self._async_last = 0

    def submit(self, iq, tv_sec=0, tv_usec=0):
        """one block of host samples per stream (numpy; memory from Context.host_alloc is used in place); returns at once.
        Raises SdrHipError(code SDRHIP_EBUSY = -6) when every batch of the ring is in flight."""
        if _is_torch(iq):
            raise TypeError("submit takes host memory")
        a = np.asarray(iq)
        if a.ndim == 3 and a.dtype == np.int16 and a.shape[0] == self.nstreams and a.shape[2] == 2 and a.strides[2] == 2 and a.strides[1] == 4 and a.strides[0] % 4 == 0:
            x = a  # rows of a bigger array (e.g. a pinned buffer): passed in place with their stride
        else:
            x, _, _ = _bank_view(iq, self.nstreams)
        if not hasattr(self, "_async_batches"):  # (sdrhip_rx_submit's default ring: 4 batches of one block)
            self._async_blocks, self._async_batches, self._async_fill, self._async_last = 1, [], [0, 0], 0
        check(self.ctx.lib.sdrhip_rx_submit(self.h, _ptr(x), x.shape[1], _stride_samples(x), tv_sec, tv_usec))
        if x.shape[1]:
            self._async_fill[0] += x.shape[1]
            self._async_fill[1] += 1
            if self._async_fill[1] >= self._async_blocks:  # (the library launched the batch)
                self._async_batches.append(self._async_fill[0])
                self._async_fill = [0, 0]

    def collect(self, wait=True, max_frames=None):
        """-> the finished frames of the oldest batch (S, n, 128 + nb_fec, 512; n may be 0), or None when no batch was collected:
        nothing submitted, or (wait = False) the oldest batch is still in flight / being filled"""
        batches = getattr(self, "_async_batches", [])
        fill = getattr(self, "_async_fill", [0, 0])
        biggest = max([getattr(self, "_async_last", 0), fill[0]] + batches[:1])  # the oldest batch, or its predecessor (pipelined)
        cap = max_frames if max_frames is not None else max(biggest // (SAMPLES_PER_FRAME << self.cfg.log2decim) + 2, 1)
        fb = (NB_ORIGINAL + self.nb_fec) * UDPSIZE
        nf = C.c_size_t(0)
        for _ in range(2):  # (a batch bigger than the guess -- pipelined mode delivers the previous batch's frames -- is asked for again)
            out = np.empty((self.nstreams, cap, NB_ORIGINAL + self.nb_fec, UDPSIZE), np.uint8)
            rc = self.ctx.lib.sdrhip_rx_collect(self.h, _ptr(out), cap * fb, cap, C.byref(nf), 1 if wait else 0)
            if rc == -1 and nf.value > cap:
                cap = nf.value
                continue
            break
        if rc == -6:
            return None
        check(rc)
        if batches:
            self._async_last = batches.pop(0)
        elif fill[1]:  # (wait = True sent the partly filled batch out as it was)
            self._async_last = fill[0]
            self._async_fill = [0, 0]
        return out[:, :nf.value]

    def process(self, iq, tv_sec=0, tv_usec=0, out=None):
        """-> frames (S, n_frames, 128 + nb_fec, 512) uint8 (squeezed for one stream)"""
        x, is_t, squeeze = _bank_view(iq, self.nstreams)
        S, n = x.shape[0], x.shape[1]
        cap = max(self.max_frames(n), 1)
        fb = (NB_ORIGINAL + self.nb_fec) * UDPSIZE
        if out is None:
            out = (torch.empty((S, cap, NB_ORIGINAL + self.nb_fec, UDPSIZE), dtype=torch.uint8, device=x.device) if is_t
                   else np.empty((S, cap, NB_ORIGINAL + self.nb_fec, UDPSIZE), np.uint8))
        if out.shape[1] < cap or out.shape[0] != S:
            raise ValueError("out must hold (S, >= %d, %d, 512) bytes" % (cap, NB_ORIGINAL + self.nb_fec))
        stride_bytes = out.shape[1] * fb
        nf = C.c_size_t(0)
        check(self.ctx.lib.sdrhip_rx_process(self.h, _ptr(x), n, _stride_samples(x), tv_sec, tv_usec, _ptr(out), stride_bytes,
                                             C.byref(nf), MEM_DEVICE if is_t else MEM_HOST))
        out = out[:, :nf.value]
        return out[0] if squeeze else out

    def process_view(self, iq, tv_sec=0, tv_usec=0):
        """Zero-copy variant for CUDA tensors: the finished frames stay in the library's frame area.
        -> uint8 CUDA tensor view (S, n_frames, 128 + nb_fec, 512), valid until the next process call."""
        x, is_t, squeeze = _bank_view(iq, self.nstreams)
        if not is_t:
            raise TypeError("process_view needs a CUDA tensor")
        S, n = x.shape[0], x.shape[1]
        nf = C.c_size_t(0)
        check(self.ctx.lib.sdrhip_rx_process(self.h, _ptr(x), n, _stride_samples(x), tv_sec, tv_usec, C.c_void_p(0), 0,
                                             C.byref(nf), MEM_DEVICE))
        return self._view(x.device)

    def close(self):
        if self.h:
            self.ctx.lib.sdrhip_rx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TxPipe:
    """SDRdaemonFECBuffer decode -> Upsampler for a bank of streams (sdrhip_tx).  pipelined=True: process() decodes its batch
    on the context's second stream while the previous batch is interpolated, and returns the PREVIOUS batch's samples
    (sdrhip_tx_set_pipelined); flush() returns the last batch's at the end.  A device-memory rx batch must stay untouched until
    the next process() / flush() has returned."""

    def __init__(self, ctx, nstreams=1, log2interp=4, pipelined=False):
        self.ctx, self.nstreams, self.log2interp = ctx, nstreams, log2interp
        self.h = C.c_void_p()
        self.m_error = ""
        self.pipelined = bool(pipelined)
        check(ctx.lib.sdrhip_tx_create(ctx.h, nstreams, log2interp, C.byref(self.h)))
        if pipelined:
            check(ctx.lib.sdrhip_tx_set_pipelined(self.h, 1))

    def configure(self, m):
        """The `interp` key of a control message (Upsampler::configure, Upsampler.cpp:31-50) between two batches;
        -> bool, the message is kept for error()."""
        if "interp" in m:
            try:
                check(self.ctx.lib.sdrhip_tx_reconfigure(self.h, int(m["interp"])))
                self.log2interp = int(m["interp"])
            except (ValueError, SdrHipError) as e:
                self.m_error = str(e)
                return False
        return True

    def error(self):
        e, self.m_error = self.m_error, ""
        return e

    def process(self, rx, indices=None):
        """rx (S, F, 128, 512) uint8 (or (F, 128, 512)) -> iq (S, F*16129 << log2interp, 2) int16"""
        is_t = _is_torch(rx)
        squeeze = rx.ndim == 3
        if squeeze:
            rx = rx[None]
        S, F = rx.shape[0], rx.shape[1]
        if not is_t:
            rx = np.ascontiguousarray(rx, dtype=np.uint8)
        else:
            rx = rx.contiguous()
        if indices is not None:  # optional (see fec_decode_frames)
            indices = np.ascontiguousarray(indices, dtype=np.uint8)
        n_res = (F * SAMPLES_PER_FRAME) << self.log2interp
        if self.pipelined:  # (the call delivers the batch the PREVIOUS call decoded)
            n_res = self.ctx.lib.sdrhip_tx_pending_samples(self.h)
        pad = max((n_res + 3) & ~3, 4)
        out = (torch.empty((S, pad, 2), dtype=torch.int16, device=rx.device) if is_t else np.empty((S, pad, 2), np.int16))
        n_out = C.c_size_t(0)
        check(self.ctx.lib.sdrhip_tx_process(self.h, _ptr(rx), C.c_void_p(indices.ctypes.data if indices is not None else 0), F, F * NB_ORIGINAL * UDPSIZE,
                                             _ptr(out), pad, C.byref(n_out), MEM_DEVICE if is_t else MEM_HOST))
        out = out[:, :n_out.value if self.pipelined else n_res]
        return out[0] if squeeze else out

    # ---- asynchronous host-pointer entry (sdrhip_tx_submit / sdrhip_tx_collect)
    def set_async(self, depth=4):
        """ring of `depth` batches in flight"""
        check(self.ctx.lib.sdrhip_tx_set_async(self.h, depth))

    def submit(self, rx, indices=None):
        """one batch of received frames from host memory, (S, F, 128, 512) uint8 (or (F, 128, 512)); returns at once.  Raises
        SdrHipError(code SDRHIP_EBUSY = -6) when every batch of the ring is in flight."""
        if _is_torch(rx):
            raise TypeError("submit takes host memory")
        a = np.asarray(rx)
        if a.ndim == 3:
            a = a[None]
        if a.dtype != np.uint8 or a.ndim != 4 or a.shape[0] != self.nstreams or a.shape[2:] != (NB_ORIGINAL, UDPSIZE):
            raise ValueError("expected (%d, F, 128, 512) uint8" % self.nstreams)
        if not (a.strides[3] == 1 and a.strides[2] == UDPSIZE and a.strides[1] == NB_ORIGINAL * UDPSIZE):
            a = np.ascontiguousarray(a)
        if indices is not None:
            indices = np.ascontiguousarray(indices, dtype=np.uint8)
        self._async_frames = getattr(self, "_async_frames", [])
        check(self.ctx.lib.sdrhip_tx_submit(self.h, _ptr(a), C.c_void_p(indices.ctypes.data if indices is not None else 0), a.shape[1], a.strides[0]))
        if a.shape[1]:
            self._async_frames.append((a.shape[1], self.log2interp))

    def collect(self, wait=True, block0=False):
        """-> the samples of the oldest batch (S, F * 16129 << log2interp, 2) int16 -- with block0=True a pair (samples, meta blocks
        (S, F, 508) uint8) -- or None when no batch was collected (nothing submitted, or wait=False and the oldest one is in flight)"""
        pend = getattr(self, "_async_frames", [])
        F, L = pend[0] if pend else (0, self.log2interp)
        cap = max((F * SAMPLES_PER_FRAME) << L, 4)
        out = np.empty((self.nstreams, cap, 2), np.int16)
        b0 = np.empty((self.nstreams, max(F, 1), BLOCK_BYTES), np.uint8)
        n_out, nf = C.c_size_t(0), C.c_size_t(0)
        rc = self.ctx.lib.sdrhip_tx_collect(self.h, _ptr(out), cap, cap, _ptr(b0) if block0 else C.c_void_p(0), C.byref(n_out), C.byref(nf), 1 if wait else 0)
        if rc == -6:
            return None
        check(rc)
        if pend:
            pend.pop(0)
        return (out[:, :n_out.value], b0[:, :nf.value]) if block0 else out[:, :n_out.value]

    def flush(self, device=None):
        """pipelined mode: the samples of the batch the last process() call decoded (sdrhip_tx_flush); (S, 0, 2) when nothing
        waits.  device: a torch device for a device-memory result, None for numpy"""
        S = self.nstreams
        n_res = self.ctx.lib.sdrhip_tx_pending_samples(self.h)
        pad = max((n_res + 3) & ~3, 4)
        out = torch.empty((S, pad, 2), dtype=torch.int16, device=device) if device is not None else np.empty((S, pad, 2), np.int16)
        n_out = C.c_size_t(0)
        check(self.ctx.lib.sdrhip_tx_flush(self.h, _ptr(out), pad, C.byref(n_out), MEM_DEVICE if device is not None else MEM_HOST))
        return out[:, :n_out.value]

    def close(self):
        if self.h:
            self.ctx.lib.sdrhip_tx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
