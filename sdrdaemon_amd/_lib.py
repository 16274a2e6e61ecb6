"""ctypes binding of libsdrhip.so (the C ABI of include/sdrhip.h).

The library is the product: there is no Python or CPU fallback.  If the shared object is
missing the import fails loudly; if no GPU is present every compute call fails with
SDRHIP_EDEVICE.  torch is imported first so that libsdrhip.so binds to the HIP runtime
torch already loaded (same SONAME libamdhip64.so.7) and device pointers of torch tensors
are valid in its launches.
"""
import ctypes as C
import os

try:  # plumbing only: shared HIP runtime + device tensors
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for pure host-pointer use
    torch = None

HERE = os.path.dirname(os.path.abspath(__file__))
# SDRHIP_LIB_PATH: a variant build of the same library (tools/experiments_*: kernel A / B runs); there is still no fallback
LIB_PATH = os.environ.get("SDRHIP_LIB_PATH") or os.path.join(HERE, "libsdrhip.so")

MEM_HOST, MEM_DEVICE = 0, 1
FC_INF, FC_SUP, FC_CEN = 0, 1, 2
HB_EO1, HB_DB = 0, 1
UDPSIZE, NB_ORIGINAL, BLOCK_BYTES, SAMPLES_PER_BLOCK, SAMPLES_PER_FRAME = 512, 128, 508, 127, 16129

EXPORTS = [
    "sdrhip_last_error", "sdrhip_device_count", "sdrhip_ctx_create", "sdrhip_ctx_destroy", "sdrhip_ctx_synchronize", "sdrhip_ctx_set_option", "sdrhip_ctx_get_counter", "sdrhip_decimators_last_plan", "sdrhip_rx_last_plan", "sdrhip_rx_set_pipelined", "sdrhip_rx_flush", "sdrhip_rx_set_async", "sdrhip_rx_submit", "sdrhip_rx_collect", "sdrhip_host_alloc", "sdrhip_host_free",
    "sdrhip_ctx_timing_begin", "sdrhip_ctx_timing_end", "sdrhip_ctx_kernel_timing", "sdrhip_ctx_kernel_timing_read", "sdrhip_decimators_create", "sdrhip_decimators_destroy",
    "sdrhip_decimators_reset", "sdrhip_decimate", "sdrhip_interpolators_create", "sdrhip_interpolators_destroy",
    "sdrhip_interpolators_reset", "sdrhip_interpolate", "sdrhip_cm256_encode", "sdrhip_cm256_decode",
    "sdrhip_fec_encode_frames", "sdrhip_fec_decode_frames", "sdrhip_rx_create", "sdrhip_rx_destroy", "sdrhip_rx_reconfigure", "sdrhip_rx_process",
    "sdrhip_rx_max_frames", "sdrhip_rx_frames_view", "sdrhip_tx_create", "sdrhip_tx_destroy", "sdrhip_tx_reconfigure", "sdrhip_tx_process", "sdrhip_tx_set_pipelined", "sdrhip_tx_flush", "sdrhip_tx_pending_samples", "sdrhip_tx_set_async", "sdrhip_tx_submit", "sdrhip_tx_collect",
    "sdrhip_testsource_create", "sdrhip_testsource_destroy", "sdrhip_testsource_configure", "sdrhip_testsource_get", "sdrhip_testsource_read",
]


class SdrHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sdrhip error %d: %s" % (code, msg))
        self.code = code


class CM256Params(C.Structure):
    _fields_ = [("OriginalCount", C.c_int), ("RecoveryCount", C.c_int), ("BlockBytes", C.c_int)]


class CM256Block(C.Structure):
    _fields_ = [("Block", C.c_void_p), ("Index", C.c_ubyte)]


class DecimPlan(C.Structure):
    _fields_ = [("path", C.c_int), ("wps", C.c_int), ("npieces", C.c_int), ("nseg", C.c_int),
                ("span", C.c_size_t), ("head", C.c_size_t), ("tail_start", C.c_size_t)]


class RxConfig(C.Structure):
    _fields_ = [("log2decim", C.c_int), ("fcpos", C.c_int), ("hb_variant", C.c_int), ("sample_bits", C.c_uint),
                ("nb_fec", C.c_int), ("center_frequency_khz", C.c_uint32), ("sample_rate", C.c_uint32)]


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, sz, i, u = C.c_void_p, C.c_size_t, C.c_int, C.c_uint
    lib.sdrhip_last_error.restype = C.c_char_p
    lib.sdrhip_device_count.restype = i
    lib.sdrhip_ctx_create.argtypes = [i, vp, C.POINTER(vp)]
    lib.sdrhip_ctx_destroy.argtypes = [vp]
    lib.sdrhip_ctx_destroy.restype = None
    lib.sdrhip_ctx_synchronize.argtypes = [vp]
    lib.sdrhip_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.sdrhip_ctx_get_counter.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64)]
    lib.sdrhip_decimators_last_plan.argtypes = [vp, C.POINTER(DecimPlan)]
    lib.sdrhip_rx_last_plan.argtypes = [vp, C.POINTER(DecimPlan)]
    lib.sdrhip_ctx_timing_begin.argtypes = [vp]
    lib.sdrhip_ctx_timing_end.argtypes = [vp, C.POINTER(C.c_float)]
    lib.sdrhip_ctx_kernel_timing.argtypes = [vp, i]
    lib.sdrhip_ctx_kernel_timing_read.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(u)]
    lib.sdrhip_decimators_create.argtypes = [vp, i, i, C.POINTER(vp)]
    lib.sdrhip_decimators_destroy.argtypes = [vp]
    lib.sdrhip_decimators_destroy.restype = None
    lib.sdrhip_decimators_reset.argtypes = [vp]
    lib.sdrhip_decimate.argtypes = [vp, i, i, C.POINTER(u), vp, sz, sz, vp, sz, C.POINTER(sz), i]
    lib.sdrhip_interpolators_create.argtypes = [vp, i, C.POINTER(vp)]
    lib.sdrhip_interpolators_destroy.argtypes = [vp]
    lib.sdrhip_interpolators_destroy.restype = None
    lib.sdrhip_interpolators_reset.argtypes = [vp]
    lib.sdrhip_interpolate.argtypes = [vp, i, vp, sz, sz, vp, sz, C.POINTER(sz), i]
    lib.sdrhip_cm256_encode.argtypes = [vp, CM256Params, C.POINTER(CM256Block), vp]
    lib.sdrhip_cm256_decode.argtypes = [vp, CM256Params, C.POINTER(CM256Block)]
    lib.sdrhip_fec_encode_frames.argtypes = [vp, vp, sz, i, vp, i]
    lib.sdrhip_fec_decode_frames.argtypes = [vp, vp, vp, sz, vp, vp, i]
    lib.sdrhip_rx_create.argtypes = [vp, i, C.POINTER(RxConfig), C.POINTER(vp)]
    lib.sdrhip_rx_destroy.argtypes = [vp]
    lib.sdrhip_rx_reconfigure.argtypes = [vp, C.POINTER(RxConfig)]
    lib.sdrhip_rx_destroy.restype = None
    lib.sdrhip_rx_process.argtypes = [vp, vp, sz, sz, C.c_uint32, C.c_uint32, vp, sz, C.POINTER(sz), i]
    lib.sdrhip_rx_frames_view.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    lib.sdrhip_rx_set_pipelined.argtypes = [vp, i]
    lib.sdrhip_rx_flush.argtypes = [vp, vp, sz, C.POINTER(sz), i]
    lib.sdrhip_rx_set_async.argtypes = [vp, i, i]
    lib.sdrhip_rx_submit.argtypes = [vp, vp, sz, sz, C.c_uint32, C.c_uint32]
    lib.sdrhip_rx_collect.argtypes = [vp, vp, sz, sz, C.POINTER(sz), i]
    lib.sdrhip_host_alloc.argtypes = [vp, sz]
    lib.sdrhip_host_alloc.restype = vp
    lib.sdrhip_host_free.argtypes = [vp, vp]
    lib.sdrhip_host_free.restype = None
    lib.sdrhip_rx_max_frames.argtypes = [vp, sz]
    lib.sdrhip_rx_max_frames.restype = sz
    lib.sdrhip_tx_create.argtypes = [vp, i, i, C.POINTER(vp)]
    lib.sdrhip_tx_reconfigure.argtypes = [vp, i]
    lib.sdrhip_tx_destroy.argtypes = [vp]
    lib.sdrhip_tx_destroy.restype = None
    lib.sdrhip_tx_process.argtypes = [vp, vp, vp, sz, sz, vp, sz, C.POINTER(sz), i]
    lib.sdrhip_tx_set_pipelined.argtypes = [vp, i]
    lib.sdrhip_tx_flush.argtypes = [vp, vp, sz, C.POINTER(sz), i]
    lib.sdrhip_tx_pending_samples.argtypes = [vp]
    lib.sdrhip_tx_pending_samples.restype = sz
    lib.sdrhip_tx_set_async.argtypes = [vp, i]
    lib.sdrhip_tx_submit.argtypes = [vp, vp, vp, sz, sz]
    lib.sdrhip_tx_collect.argtypes = [vp, vp, sz, sz, vp, C.POINTER(sz), C.POINTER(sz), i]
    lib.sdrhip_testsource_create.argtypes = [vp, i, C.POINTER(vp)]
    lib.sdrhip_testsource_destroy.argtypes = [vp]
    lib.sdrhip_testsource_destroy.restype = None
    lib.sdrhip_testsource_configure.argtypes = [vp, i, C.c_char_p]
    lib.sdrhip_testsource_get.argtypes = [vp, i, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    lib.sdrhip_testsource_read.argtypes = [vp, vp, sz, sz, i]
    return lib


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = load()
    return _LIB


def check(rc):
    if rc != 0:
        raise SdrHipError(rc, lib().sdrhip_last_error().decode("utf-8", "replace"))
