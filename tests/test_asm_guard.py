"""Build-time guard: the CM256 walks start their LDS table loads in one asm statement and await them in a later one; nothing
may touch the destination registers in between (the hardware does not interlock them).  tools/check_asm_tables.py compiles the two
kernel files to assembly and proves it for every kernel.  CPU only (hipcc cross-compiles), ~50 s."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_no_instruction_touches_a_table_register_before_its_wait():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_tables.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 violations" in r.stdout and "asm-issued LDS table loads" in r.stdout, r.stdout
