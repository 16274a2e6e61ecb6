"""The reference's OWN SDRdaemonFECBuffer.cpp, compiled where it lies (tests/cxx/Makefile) against the product's
drop-in cm256.h adapter = libsdrhip.so: the reference's decoder call site (SDRdaemonFECBuffer.cpp:148-213: block
collection, first-128 policy, decode call, fix-up loop, frame change emission, stats) runs unchanged on the GPU and
gives what the oracle's restatement of the same chain gives.  (Round 1 also built that class over the ORACLE's CM256
through a stand-in cm256.h; that build is gone: no reference code is compiled against headers of our own making
except the adapter that IS the product's boundary.)"""
import ctypes as C
import os

import numpy as np
import pytest

import signals


def _load(name):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cxx", "_build", name)
    if not os.path.exists(path):
        pytest.skip("%s not built" % name)
    L = C.CDLL(path)
    L.sdrref_fecbuf_new.restype = C.c_void_p
    L.sdrref_fecbuf_free.argtypes = [C.c_void_p]
    L.sdrref_fecbuf_write_and_read.restype = C.c_int
    L.sdrref_fecbuf_write_and_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    for f in ("cur_nb_blocks", "cur_nb_recovery", "min_nb_blocks", "max_nb_recovery"):
        getattr(L, "sdrref_fecbuf_" + f).restype = C.c_int
        getattr(L, "sdrref_fecbuf_" + f).argtypes = [C.c_void_p]
    return L


def _datagrams(oracle, seed, nframes=5, R=32):
    """frames with losses, in wire order (originals first, then recovery blocks: UDPSinkFEC.cpp:259-282)"""
    rs = np.random.RandomState(seed)
    x = signals.mixed(nframes * 16129, seed)
    frames = oracle.framer(nb_fec_blocks=R).write(x)
    out = []
    for f in range(nframes):
        allb = np.concatenate([frames[f], oracle.frame_encode(frames[f], R)])
        if f == 0:
            lost = set()
        elif f == 1:
            lost = set(range(1, 121, 5))           # 24 originals
        elif f == 2:
            lost = set(rs.choice(160, 30, replace=False).tolist()) | {0}
        elif f == 3:
            lost = {7}                              # one loss: the RecoveryCount == 1 shortcut, row 128 arrives first
        else:
            lost = set(rs.choice(128, 40, replace=False).tolist())  # too many: frame stays incomplete
        out += [allb[i] for i in range(160) if i not in lost]
    out.append(np.full(512, 0xEE, np.uint8))        # a datagram of the next frame flushes the last one
    return x, out


def _run_ref(L, dgrams):
    h = C.c_void_p(L.sdrref_fecbuf_new())
    outs, stats = [], []
    data = np.zeros(127 * 508, np.uint8)
    ln = C.c_size_t(0)
    for d in dgrams:
        d = np.ascontiguousarray(d)
        if L.sdrref_fecbuf_write_and_read(h, d.ctypes.data, data.ctypes.data, C.byref(ln)):
            assert ln.value == 127 * 508
            outs.append(data.copy())
            stats.append((L.sdrref_fecbuf_cur_nb_blocks(h), L.sdrref_fecbuf_cur_nb_recovery(h)))
    mm = (L.sdrref_fecbuf_min_nb_blocks(h), L.sdrref_fecbuf_max_nb_recovery(h))
    L.sdrref_fecbuf_free(h)
    return outs, stats, mm


def _run_oracle(oracle, dgrams):
    b = oracle.fecbuffer()
    outs, stats = [], []
    for d in dgrams:
        o = b.write_and_read(d)
        if o is not None:
            outs.append(o.copy())
            stats.append((b.s.cur_nb_blocks, b.s.cur_nb_recovery))
    return outs, stats, (b.s.min_nb_blocks, b.s.max_nb_recovery)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 4])
def test_reference_decoder_call_site_runs_on_the_product(oracle, seed):
    import sdrdaemon_amd as sd

    assert sd.device_count() > 0
    L = _load("libsdrref_fecbuf_hip.so")
    x, dg = _datagrams(oracle, seed)
    ro, rstats, _ = _run_ref(L, dg)
    oo, ostats, _ = _run_oracle(oracle, dg)
    assert len(ro) == len(oo) == 6
    for i in range(1, 6):
        assert np.array_equal(ro[i], oo[i]), i
    assert rstats[1:] == ostats[1:]
