"""ctypes access to the test-only CPU checker under oracle/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  `Oracle` wraps oracle/liborc.so (the C restatement), `Reference` wraps
oracle/_ref/libsdrref_{eo1,db}.so (the real reference DSP classes compiled by
oracle/Makefile in the build container; prebuilt files travel to the GPU box).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

FC_INF, FC_SUP, FC_CEN = 0, 1, 2


def build_oracle():
    """(Re)build liborc.so and, where the reference tree exists, oracle/_ref."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def _i16(a):
    a = np.ascontiguousarray(a, dtype=np.int16)
    return a, a.ctypes.data_as(C.POINTER(C.c_int16))


class CM256Block(C.Structure):
    _fields_ = [("Block", C.c_void_p), ("Index", C.c_uint8)]


class CM256Params(C.Structure):
    _fields_ = [("OriginalCount", C.c_int), ("RecoveryCount", C.c_int), ("BlockBytes", C.c_int)]


class Oracle:
    def __init__(self):
        path = os.path.join(ORACLE_DIR, "liborc.so")
        if not os.path.exists(path):
            build_oracle()
        L = self.lib = C.CDLL(path)
        L.orc_decimators_new.restype = C.c_void_p
        L.orc_decimators_new.argtypes = [C.c_int]
        L.orc_decimators_free.argtypes = [C.c_void_p]
        L.orc_decimate.restype = C.c_size_t
        L.orc_decimate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_int16),
                                   C.c_size_t, C.POINTER(C.c_int16)]
        L.orc_interpolators_new.restype = C.c_void_p
        L.orc_interpolators_free.argtypes = [C.c_void_p]
        L.orc_interpolate.restype = C.c_size_t
        L.orc_interpolate.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int16), C.c_size_t, C.POINTER(C.c_int16)]
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_gf_mul.restype = C.c_uint8
        L.orc_gf_mul.argtypes = [C.c_uint8, C.c_uint8]
        L.orc_gf_div.restype = C.c_uint8
        L.orc_gf_div.argtypes = [C.c_uint8, C.c_uint8]
        L.orc_gf_exp.restype = C.c_uint8
        L.orc_gf_exp.argtypes = [C.c_int]
        L.orc_gf_log.restype = C.c_int
        L.orc_gf_log.argtypes = [C.c_uint8]
        L.orc_cm256_matrix_element.restype = C.c_uint8
        L.orc_cm256_matrix_element.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8]
        L.orc_cm256_encode.restype = C.c_int
        L.orc_cm256_encode.argtypes = [CM256Params, C.POINTER(CM256Block), C.c_void_p]
        L.orc_cm256_decode.restype = C.c_int
        L.orc_cm256_decode.argtypes = [CM256Params, C.POINTER(CM256Block)]
        L.orc_gf_muladd_mem.argtypes = [C.c_void_p, C.c_uint8, C.c_void_p, C.c_size_t]
        L.orc_framer_init.argtypes = [C.c_void_p]
        L.orc_framer_write.restype = C.c_size_t
        L.orc_framer_write.argtypes = [C.c_void_p, C.POINTER(C.c_int16), C.c_size_t, C.c_void_p]
        L.orc_frame_encode.restype = C.c_int
        L.orc_frame_encode.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_nco_cos_q30.argtypes = [C.c_uint]
        L.orc_nco_phase_inc.argtypes = [C.c_int64, C.c_int64]
        L.orc_nco_phase_inc.restype = C.c_uint
        L.orc_nco_amp_q15.argtypes = [C.c_int]
        L.orc_testsource_generate.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_size_t, C.c_void_p]
        L.orc_testsource_generate.restype = C.c_uint
        L.orc_fecbuffer_init.argtypes = [C.c_void_p]
        L.orc_fecbuffer_write_and_read.restype = C.c_int
        L.orc_fecbuffer_write_and_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]

    # ---- DSP
    def decimators(self, bias=0):
        return OracleDecimators(self, bias)

    def interpolators(self):
        return OracleInterpolators(self)

    # ---- GF / CM256
    def gf_mul(self, a, b):
        return self.lib.orc_gf_mul(a, b)

    def gf_div(self, a, b):
        return self.lib.orc_gf_div(a, b)

    def matrix_element(self, x_i, x_0, y_j):
        return self.lib.orc_cm256_matrix_element(x_i, x_0, y_j)

    def crc32(self, data: bytes):
        return self.lib.orc_crc32(data, len(data))

    def cm256_encode(self, originals, recovery_count):
        """originals: (k, bb) uint8 -> (recovery_count, bb) uint8"""
        originals = np.ascontiguousarray(originals, dtype=np.uint8)
        k, bb = originals.shape
        blocks = (CM256Block * k)()
        for i in range(k):
            blocks[i].Block = originals[i].ctypes.data
            blocks[i].Index = i
        rec = np.zeros((recovery_count, bb), dtype=np.uint8)
        rc = self.lib.orc_cm256_encode(CM256Params(k, recovery_count, bb), blocks, rec.ctypes.data)
        if rc:
            raise RuntimeError("orc_cm256_encode failed: %d" % rc)
        return rec

    def cm256_decode(self, data, indices, original_count, recovery_count):
        """data: (k, bb) uint8 received blocks (modified in place), indices: their block indices.
        Returns (rc, indices_after) with the upstream in-place contract."""
        assert data.dtype == np.uint8 and data.flags.c_contiguous
        k, bb = data.shape
        blocks = (CM256Block * k)()
        for i in range(k):
            blocks[i].Block = data[i].ctypes.data
            blocks[i].Index = int(indices[i])
        rc = self.lib.orc_cm256_decode(CM256Params(original_count, recovery_count, bb), blocks)
        return rc, np.array([blocks[i].Index for i in range(k)], dtype=np.uint8)

    def testsource_generate(self, phase0, inc, amp_q15, n):
        """-> ((n, 2) int16, phase after the last sample): the integer NCO the product's TestSource bank defines"""
        out = np.zeros((n, 2), np.int16)
        ph = self.lib.orc_testsource_generate(phase0, inc, amp_q15, n, out.ctypes.data)
        return out, ph

    def framer(self, **kw):
        return OracleFramer(self, **kw)

    def frame_encode(self, frame, nb_fec):
        frame = np.ascontiguousarray(frame, dtype=np.uint8).reshape(128, 512)
        out = np.zeros((nb_fec, 512), dtype=np.uint8)
        rc = self.lib.orc_frame_encode(frame.ctypes.data, nb_fec, out.ctypes.data)
        if rc:
            raise RuntimeError("orc_frame_encode failed: %d" % rc)
        return out

    def fecbuffer(self):
        return OracleFECBuffer(self)


class OracleDecimators:
    def __init__(self, orc, bias):
        self.o = orc
        self.h = C.c_void_p(orc.lib.orc_decimators_new(bias))

    def __del__(self):
        if getattr(self, "h", None):
            self.o.lib.orc_decimators_free(self.h)
            self.h = None

    def decimate(self, log2decim, fcpos, sample_size, iq):
        """iq: (n, 2) int16.  Returns (out (n >> log2, 2) int16, new sample_size)."""
        a, p = _i16(iq)
        n = a.shape[0]
        out = np.zeros((max(n >> log2decim, 0), 2), dtype=np.int16)
        ss = C.c_uint(sample_size)
        no = self.o.lib.orc_decimate(self.h, log2decim, fcpos, C.byref(ss), p, n,
                                     out.ctypes.data_as(C.POINTER(C.c_int16)))
        return out[:no], ss.value


class OracleInterpolators:
    def __init__(self, orc):
        self.o = orc
        self.h = C.c_void_p(orc.lib.orc_interpolators_new())

    def __del__(self):
        if getattr(self, "h", None):
            self.o.lib.orc_interpolators_free(self.h)
            self.h = None

    def interpolate(self, log2interp, iq):
        a, p = _i16(iq)
        n = a.shape[0]
        out = np.zeros((n << log2interp, 2), dtype=np.int16)
        no = self.o.lib.orc_interpolate(self.h, log2interp, p, n, out.ctypes.data_as(C.POINTER(C.c_int16)))
        return out[:no]


class _FramerStruct(C.Structure):
    _fields_ = [("cur", C.c_uint8 * 512), ("slot", C.c_uint8 * (128 * 512)), ("tx_block_index", C.c_int),
                ("sample_index", C.c_int), ("frame_count", C.c_uint16), ("center_frequency_khz", C.c_uint32),
                ("sample_rate", C.c_uint32), ("sample_bytes", C.c_uint8), ("sample_bits", C.c_uint8),
                ("nb_fec_blocks", C.c_uint8), ("tv_sec", C.c_uint32), ("tv_usec", C.c_uint32),
                ("stamp_from_samples", C.c_int)]


class OracleFramer:
    def __init__(self, orc, center_frequency_khz=435000, sample_rate=625000, sample_bytes=2, sample_bits=16,
                 nb_fec_blocks=32, tv_sec=0, tv_usec=0, stamp_from_samples=1):
        """stamp_from_samples=1 (the product's rule): (tv_sec, tv_usec) is the time of a write() call's first sample, the frames the
        call opens are stamped by the sample clock from there; 0: the reference's literal behaviour with the test playing
        gettimeofday (every frame a call opens carries tv_sec / tv_usec as they are)."""
        self.o = orc
        self.s = _FramerStruct()
        orc.lib.orc_framer_init(C.byref(self.s))
        self.s.center_frequency_khz = center_frequency_khz
        self.s.sample_rate = sample_rate
        self.s.sample_bytes = sample_bytes
        self.s.sample_bits = sample_bits
        self.s.nb_fec_blocks = nb_fec_blocks
        self.s.tv_sec = tv_sec
        self.s.tv_usec = tv_usec
        self.s.stamp_from_samples = stamp_from_samples

    def write(self, iq):
        """Returns completed frames as (n_frames, 128, 512) uint8."""
        a, p = _i16(iq)
        n = a.shape[0]
        cap = n // 16129 + 2
        out = np.zeros((cap, 128, 512), dtype=np.uint8)
        nf = self.o.lib.orc_framer_write(C.byref(self.s), p, n, out.ctypes.data)
        return out[:nf]


class _FECBufferStruct(C.Structure):
    _fields_ = [("frame", C.c_uint8 * (128 * 508)), ("recovery", C.c_uint8 * (128 * 508)),
                ("desc", CM256Block * 128), ("block_count", C.c_int), ("recovery_count", C.c_int),
                ("decoded", C.c_int), ("meta_retrieved", C.c_int), ("frame_head", C.c_int),
                ("cur_nb_blocks", C.c_int), ("cur_nb_recovery", C.c_int), ("min_nb_blocks", C.c_int),
                ("max_nb_recovery", C.c_int)]


class OracleFECBuffer:
    def __init__(self, orc):
        self.o = orc
        self.s = _FECBufferStruct()
        orc.lib.orc_fecbuffer_init(C.byref(self.s))

    def write_and_read(self, superblock):
        sb = np.ascontiguousarray(superblock, dtype=np.uint8)
        data = np.zeros(127 * 508, dtype=np.uint8)
        ln = C.c_size_t(0)
        avail = self.o.lib.orc_fecbuffer_write_and_read(C.byref(self.s), sb.ctypes.data, data.ctypes.data,
                                                        C.byref(ln))
        return (data if avail else None)


class Reference:
    """The real reference DSP classes (Decimators / Interpolators), EO1 or DB flavour."""

    def __init__(self, flavour="eo1"):
        path = os.path.join(ORACLE_DIR, "_ref", "libsdrref_%s.so" % flavour)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = self.lib = C.CDLL(path)
        L.sdrref_bias.restype = C.c_int
        L.sdrref_decimators_new.restype = C.c_void_p
        L.sdrref_decimators_free.argtypes = [C.c_void_p]
        L.sdrref_decimate.restype = C.c_size_t
        L.sdrref_decimate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_int16),
                                      C.c_size_t, C.POINTER(C.c_int16)]
        L.sdrref_interpolators_new.restype = C.c_void_p
        L.sdrref_interpolators_free.argtypes = [C.c_void_p]
        L.sdrref_interpolate.restype = C.c_size_t
        L.sdrref_interpolate.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int16), C.c_size_t,
                                         C.POINTER(C.c_int16)]
        L.sdrref_decimate_repeat.restype = C.c_size_t
        L.sdrref_decimate_repeat.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_int16),
                                             C.c_size_t, C.POINTER(C.c_int16), C.c_int]
        L.sdrref_interpolate_repeat.restype = C.c_size_t
        L.sdrref_interpolate_repeat.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int16), C.c_size_t,
                                                C.POINTER(C.c_int16), C.c_int]
        self.bias = L.sdrref_bias()

    @staticmethod
    def available(flavour="eo1"):
        return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libsdrref_%s.so" % flavour))

    def decimators(self):
        return RefDecimators(self)

    def interpolators(self):
        return RefInterpolators(self)


class RefDecimators:
    def __init__(self, ref):
        self.r = ref
        self.h = C.c_void_p(ref.lib.sdrref_decimators_new())

    def __del__(self):
        if getattr(self, "h", None):
            self.r.lib.sdrref_decimators_free(self.h)
            self.h = None

    def decimate(self, log2decim, fcpos, sample_size, iq):
        a, p = _i16(iq)
        n = a.shape[0]
        out = np.zeros((n >> log2decim, 2), dtype=np.int16)
        ss = C.c_uint(sample_size)
        no = self.r.lib.sdrref_decimate(self.h, log2decim, fcpos, C.byref(ss), p, n,
                                        out.ctypes.data_as(C.POINTER(C.c_int16)))
        return out[:no], ss.value

    def decimate_repeat(self, log2decim, fcpos, sample_size, iq, reps):
        a, p = _i16(iq)
        n = a.shape[0]
        out = np.zeros((n >> log2decim, 2), dtype=np.int16)
        self.r.lib.sdrref_decimate_repeat(self.h, log2decim, fcpos, sample_size, p, n,
                                          out.ctypes.data_as(C.POINTER(C.c_int16)), reps)
        return out


class RefInterpolators:
    def __init__(self, ref):
        self.r = ref
        self.h = C.c_void_p(ref.lib.sdrref_interpolators_new())

    def __del__(self):
        if getattr(self, "h", None):
            self.r.lib.sdrref_interpolators_free(self.h)
            self.h = None

    def interpolate(self, log2interp, iq):
        a, p = _i16(iq)
        n = a.shape[0]
        out = np.zeros((n << log2interp, 2), dtype=np.int16)
        no = self.r.lib.sdrref_interpolate(self.h, log2interp, p, n, out.ctypes.data_as(C.POINTER(C.c_int16)))
        return out[:no]

    def interpolate_repeat(self, log2interp, iq, reps):
        a, p = _i16(iq)
        n = a.shape[0]
        out = np.zeros((n << log2interp, 2), dtype=np.int16)
        self.r.lib.sdrref_interpolate_repeat(self.h, log2interp, p, n,
                                             out.ctypes.data_as(C.POINTER(C.c_int16)), reps)
        return out
