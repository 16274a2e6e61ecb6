"""CPU-side checks of the drop-in boundary: libsdrhip.so loads, exports every symbol that
include/sdrhip.h declares, and fails loudly (no CPU fallback) without a GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    from sdrdaemon_amd import _lib

    return _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "sdrhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdrhip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(built):
    lib = built.lib()
    names = _declared()
    assert len(names) >= 26
    for n in names:
        assert hasattr(lib, n), "libsdrhip.so does not export %s" % n
    assert sorted(built.EXPORTS) == names, "python binding list differs from include/sdrhip.h"


def test_header_compiles_as_c_and_cxx(tmp_path):
    import subprocess

    c = tmp_path / "t.c"
    c.write_text('#include "sdrhip.h"\nint main(void){sdrhip_cm256_params p={128,32,508};return p.BlockBytes!=508;}\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(c), "-c", "-o", str(tmp_path / "t.o")], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-x", "c++", "-I", inc, str(c), "-c", "-o", str(tmp_path / "t2.o")],
                   check=True)


def test_no_gpu_means_loud_failure(built):
    import sdrdaemon_amd as sd

    if sd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sd.SdrHipError) as e:
        sd.Context(0)
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)


def test_product_never_touches_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sdrdaemon_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_lib" not in txt and "liborc" not in txt and "sdr_oracle" not in txt, f
    import subprocess

    out = subprocess.run(["ldd", os.path.join(ROOT, "sdrdaemon_amd", "libsdrhip.so")], capture_output=True, text=True).stdout
    assert "liborc" not in out and "sdrref" not in out
