"""The C++11 drop-in adapters (sdrdaemon_amd/adapters: Decimators.h, Interpolators.h, cm256.h).
CPU: they compile stand-alone and against the reference's own SDRDaemon.h.  GPU: a program written
like the reference's call sites runs through them and matches the oracle."""
import os
import subprocess

import numpy as np
import pytest

import signals

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cxx", "adapter_test.cpp")
INC = ["-I", os.path.join(ROOT, "sdrdaemon_amd", "adapters"), "-I", os.path.join(ROOT, "include")]


def test_adapters_compile_standalone():
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only"] + INC + [SRC], check=True)


@pytest.mark.skipif(not os.path.exists("/root/reference/include/SDRDaemon.h"), reason="reference tree not present")
def test_adapters_compile_against_reference_headers():
    # the adapters directory comes first, as in the CMake stub of INTEGRATION.md: Decimators.h, Interpolators.h, cm256.h,
    # Downsampler.h, Upsampler.h are the adapters, SDRDaemon.h (IQSample) is the reference's
    for flags in (["-DUSE_SSE4_1"], []):
        subprocess.run(["g++", "-std=c++11", "-Wall", "-fsyntax-only"] + flags + INC + ["-I", "/root/reference/include", SRC], check=True)


@pytest.mark.gpu
def test_adapters_run_like_the_reference_call_sites(tmp_path, oracle):
    import __graft_entry__ as g

    g.build()
    exe = str(tmp_path / "adapter_test")
    libdir = os.path.join(ROOT, "sdrdaemon_amd")
    subprocess.run(["g++", "-std=c++11", "-O1", "-pthread"] + INC + [SRC, "-L", libdir, "-lsdrhip", "-Wl,-rpath," + libdir, "-o", exe],
                   check=True)
    x = signals.mixed(2 * 65536, 9)
    fin = str(tmp_path / "in.bin")
    x.tofile(fin)
    outs = [str(tmp_path / n) for n in ("dec.bin", "int.bin", "fec.bin", "smp.bin")]
    env = dict(os.environ, SDRHIP_HB_VARIANT="EO1")
    r = subprocess.run([exe, fin] + outs, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "sampleSize 16 14" in r.stdout and "fec roundtrip OK" in r.stdout, r.stdout
    assert "threads OK" in r.stdout, r.stdout  # main-thread decimation, transmit-thread CM256, static entry points: concurrently
    od = oracle.decimators(0)
    a, _ = od.decimate(4, 2, 16, x[:65536])
    b, _ = od.decimate(4, 2, 16, x[65536:])
    c, _ = od.decimate(3, 0, 16, x[:65536])
    d, ss4 = oracle.decimators(0).decimate(2, 1, 12, x[:65536])
    assert ss4 == 14
    exp = np.concatenate([a, b, c, d])
    got = np.fromfile(outs[0], dtype=np.int16).reshape(-1, 2)
    assert np.array_equal(got, exp)
    ou = oracle.interpolators()
    exp_i = np.concatenate([ou.interpolate(4, a[:1000]), ou.interpolate(4, a[1000:])])
    assert np.array_equal(np.fromfile(outs[1], dtype=np.int16).reshape(-1, 2), exp_i)
    orig = x.view(np.uint8).reshape(-1)[:128 * 508].reshape(128, 508)
    assert np.array_equal(np.fromfile(outs[2], dtype=np.uint8).reshape(32, 508), oracle.cm256_encode(orig, 32))
    # Downsampler / Upsampler adapters: decimate16_cen, reconfigured to decimate8_inf on the same bank, decim 0, interp 4
    od2 = oracle.decimators(0)
    s1, _ = od2.decimate(4, 2, 16, x[:65536])
    s2, _ = od2.decimate(3, 0, 16, x[65536:])
    s3, ss3 = od2.decimate(0, 2, 12, x[:65536])
    assert "samplers OK %d" % ss3 in r.stdout, r.stdout  # (sampleSize after the decim-0 call)
    s4 = oracle.interpolators().interpolate(2, s1)
    assert np.array_equal(np.fromfile(outs[3], dtype=np.int16).reshape(-1, 2), np.concatenate([s1, s2, s3, s4]))
