"""A deterministic slice of tools/fuzz_gpu.py inside `pytest -m gpu` (the fuzzer ran 74 k iterations per round by
hand and never under the driver): fixed seeds, a cap on the iterations, every entry point the fuzzer knows -- decimators and
interpolators in every mode and kernel path, the Rx pipe (immediate, pipelined with the encode fused / separate / on the second
stream, asynchronous submit / collect, live reconfiguration), the Tx pipe (immediate, pipelined on one or two streams, dec_max_rows
promises that frames break), generic CM256 geometries -- against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,iters", [(11, 150), (12, 150)])
def test_fuzz_slice(seed, iters):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "550", str(seed), str(iters)], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("fuzz OK"), last
    # the slice is iteration-bound (the 550 s budget is a backstop under the 600 s timeout, ~10 x what 150 iterations take on an
    # MI355X box): it covers the same cases on every run, whatever the box's speed
    assert ("%d iterations" % iters) in last, last
