"""The UDPSinkFEC / UDPSourceFEC drop-in adapters (sdrdaemon_amd/adapters) over the loopback interface.
CPU: they compile stand-alone and in front of the reference's include directory.
GPU: a program written like the reference's main loops (tests/cxx/udp_adapter_test.cpp) sends / receives
real datagrams; this test is the peer and checks them against the oracle chain (framer + CM256 encoder,
frame collector + CM256 decoder)."""
import os
import socket
import struct
import subprocess
import threading
import time
import zlib

import numpy as np
import pytest

import signals

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cxx", "udp_adapter_test.cpp")
INC = ["-I", os.path.join(ROOT, "sdrdaemon_amd", "adapters"), "-I", os.path.join(ROOT, "include")]


def test_udp_adapters_compile_standalone():
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only"] + INC + [SRC], check=True)


@pytest.mark.skipif(not os.path.exists("/root/reference/include/SDRDaemon.h"), reason="reference tree not present")
def test_udp_adapters_compile_in_front_of_the_reference_headers():
    # the adapters' UDPSink.h / UDPSource.h / UDPSinkFEC.h / UDPSourceFEC.h shadow the reference's, IQSample comes from SDRDaemon.h
    subprocess.run(["g++", "-std=c++11", "-Wall", "-fsyntax-only"] + INC + ["-I", "/root/reference/include", SRC], check=True)


def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import __graft_entry__ as g

    g.build()
    out = str(tmp_path_factory.mktemp("udp") / "udp_adapter_test")
    libdir = os.path.join(ROOT, "sdrdaemon_amd")
    subprocess.run(["g++", "-std=c++11", "-O1"] + INC + [SRC, "-L", libdir, "-lsdrhip", "-lpthread", "-Wl,-rpath," + libdir, "-o", out],
                   check=True)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("nb_fec,chunk", [(32, 4096), (8, 16129 + 77), (0, 1000)])
def test_udpsinkfec_datagrams(exe, oracle, tmp_path, nb_fec, chunk):
    nframes = 3
    # one more frame than will arrive: like the reference's transmit thread the adapter's sends frame i when frame i + 1 is
    # complete (UDPSinkFEC.cpp:160,206-211); the tail stays in the open frame
    x = signals.mixed((nframes + 1) * 16129 + 500, 21 + nb_fec)
    fin = str(tmp_path / "in.bin")
    x.tofile(fin)
    port = _free_port()
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 8 << 20)
    rx.bind(("127.0.0.1", port))
    rx.settimeout(0.5)
    got = []
    stop = threading.Event()

    def pump():
        while not stop.is_set():
            try:
                got.append(rx.recv(2048))
            except socket.timeout:
                pass

    th = threading.Thread(target=pump)
    th.start()
    try:
        r = subprocess.run([exe, "tx", str(port), str(nb_fec), "30", fin, str(chunk)], capture_output=True, text=True, timeout=120)
    finally:
        time.sleep(0.3)
        stop.set()
        th.join()
        rx.close()
    assert r.returncode == 0 and "tx done" in r.stdout, r.stderr
    per = 128 + nb_fec
    assert len(got) == nframes * per, (len(got), r.stderr)
    assert all(len(d) == 512 for d in got)
    dg = np.frombuffer(b"".join(got), np.uint8).reshape(nframes, per, 512)
    for f in range(nframes):
        fr = dg[f]
        # headers {frameIndex, blockIndex, 0} in sending order: originals, then recovery (UDPSinkFEC.cpp:259-282)
        assert np.array_equal(fr[:, 0] | (fr[:, 1].astype(np.uint16) << 8), np.full(per, f)), f
        assert np.array_equal(fr[:, 2], np.arange(per)), f
        assert not fr[:, 3].any()
        # block 0: MetaDataFEC (UDPSinkFEC.h:77-101) + zero fill
        fc, sr, sby, sbi, nbo, nbf, tvs, tvu, crc = struct.unpack("<IIBBBBIII", fr[0, 4:28].tobytes())
        assert (fc, sr, sby, sbi, nbo, nbf) == (435000, 625000, 2, 16, 128, nb_fec)
        assert crc == zlib.crc32(fr[0, 4:24].tobytes()) & 0xFFFFFFFF
        assert abs(tvs - time.time()) < 600 and tvu < 1000000
        assert not fr[0, 28:].any()
        # blocks 1..127: the stream, 127 samples each
        pay = fr[1:128, 4:].reshape(-1).view(np.int16).reshape(-1, 2)
        assert np.array_equal(pay, x[f * 16129:(f + 1) * 16129]), f
        # recovery blocks = CM256 rows 128.. over the 128 super blocks as sent (meta block included)
        if nb_fec:
            assert np.array_equal(fr[128:], oracle.frame_encode(fr[:128], nb_fec)), f


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 5])
def test_udpsourcefec_frames(exe, oracle, tmp_path, seed):
    from test_ref_fecbuffer import _datagrams, _run_oracle

    x, dgrams = _datagrams(oracle, seed)  # 5 frames with losses (none, 24, 30 incl. block 0, 1, too many) + a flush datagram
    exp, _, _ = _run_oracle(oracle, dgrams)
    assert len(exp) == 6
    port = _free_port()
    fout = str(tmp_path / "out.bin")
    p = subprocess.Popen([exe, "rx", str(port), "6", fout], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        assert p.stdout.readline().strip() == "ready"
        tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        for d in dgrams:
            tx.sendto(np.ascontiguousarray(d).tobytes(), ("127.0.0.1", port))
            time.sleep(0.0005)  # the loopback receive buffer is small; the adapter decodes between datagrams
        tx.sendto(b"short", ("127.0.0.1", port))  # wrong length: ignored (UDPSourceFEC.cpp:63)
        out, err = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 0, err
    got = np.fromfile(fout, np.uint8).reshape(6, 127 * 508)
    assert not got[0].any()  # the collector's initial slot (the reference emits uninitialised memory here)
    for i in range(1, 6):
        assert np.array_equal(got[i], exp[i]), i
    for f in range(4):  # the four decodable frames carry the stream
        assert np.array_equal(got[f + 1].view(np.int16).reshape(-1, 2), x[f * 16129:(f + 1) * 16129]), f
    lines = [ln for ln in out.splitlines() if ln.startswith("frame ")]
    assert len(lines) == 6 and all("samples 16129" in ln and "bytes 1 bits 8" in ln for ln in lines)
    # status ":<code>:<min blocks>/<max recovery>" of the frame just released (UDPSourceFEC.cpp:80-95)
    assert "status st:2:160/000" in lines[1]      # frame 0: all 160 blocks arrived, none needed
    assert "status st:0:136/024" in lines[2]      # frame 1: 24 originals lost, 136 blocks arrived
    assert "/001" in lines[4]                     # frame 3: one recovery block used
    assert "status st:1:120/" in lines[5]         # frame 4: 40 originals lost of 160 -> 120 < 128: data lost


def _no_gpu_env():
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""  # the adapters must fall back to "no FEC" like the reference without a valid CM256
    env["ROCR_VISIBLE_DEVICES"] = ""
    return env


def test_udpsinkfec_without_gpu_sends_the_originals(exe, tmp_path):
    """CPU: no device -> no recovery blocks (the reference's `!cm256Valid` branch, UDPSinkFEC.cpp:218-225); framing,
    meta block, CRC, pacing thread and sockets still have to be right."""
    nframes = 2
    x = signals.mixed((nframes + 1) * 16129 + 17, 3)  # (a frame is sent when the next one is complete, like the reference's)
    fin = str(tmp_path / "in.bin")
    x.tofile(fin)
    port = _free_port()
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 8 << 20)
    rx.bind(("127.0.0.1", port))
    rx.settimeout(0.5)
    got, stop = [], threading.Event()

    def pump():
        while not stop.is_set():
            try:
                got.append(rx.recv(2048))
            except socket.timeout:
                pass

    th = threading.Thread(target=pump)
    th.start()
    try:
        r = subprocess.run([exe, "tx", str(port), "32", "50", fin, "5000"], capture_output=True, text=True, timeout=120, env=_no_gpu_env())
    finally:
        time.sleep(0.3)
        stop.set()
        th.join()
        rx.close()
    assert r.returncode == 0 and "tx done" in r.stdout, r.stderr
    assert len(got) == nframes * 128, (len(got), r.stderr)
    dg = np.frombuffer(b"".join(got), np.uint8).reshape(nframes, 128, 512)
    for f in range(nframes):
        assert np.array_equal(dg[f, :, 2], np.arange(128))
        fc, sr, sby, sbi, nbo, nbf, tvs, tvu, crc = struct.unpack("<IIBBBBIII", dg[f, 0, 4:28].tobytes())
        assert (fc, sr, sby, sbi, nbo, nbf) == (435000, 625000, 2, 16, 128, 32)  # the meta block still announces fecblk
        assert crc == zlib.crc32(dg[f, 0, 4:24].tobytes()) & 0xFFFFFFFF
        assert np.array_equal(dg[f, 1:, 4:].reshape(-1).view(np.int16).reshape(-1, 2), x[f * 16129:(f + 1) * 16129])


def test_udpsourcefec_without_gpu_passes_complete_frames(exe, oracle, tmp_path):
    """CPU: frames that arrive complete need no decoder: the collector has to hand them through."""
    x = signals.mixed(3 * 16129, 8)
    frames = oracle.framer(nb_fec_blocks=0).write(x)
    port = _free_port()
    fout = str(tmp_path / "out.bin")
    p = subprocess.Popen([exe, "rx", str(port), "4", fout], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_no_gpu_env())
    try:
        assert p.stdout.readline().strip() == "ready"
        tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        for f in range(3):
            for b in range(128):
                tx.sendto(frames[f, b].tobytes(), ("127.0.0.1", port))
                time.sleep(0.0002)
        tx.sendto(bytes([0xEE]) * 512, ("127.0.0.1", port))
        out, err = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 0, err
    got = np.fromfile(fout, np.uint8).reshape(4, 127 * 508)
    assert np.array_equal(got[1:].reshape(-1).view(np.int16).reshape(-1, 2), x)
