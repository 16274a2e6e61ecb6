"""Loader for tests/golden/dsp_golden.{npz,json} (made by tests/golden/make_golden.py from the
compiled reference)."""
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name="dsp_golden"):
        """name: dsp_golden (all entry points, 16384 samples) or dsp_golden_long (centred cascades, 65536 samples: long enough
        for the matrix-core kernel at every ratio)"""
        self.arrays = np.load(os.path.join(HERE, name + ".npz"))
        with open(os.path.join(HERE, name + ".json")) as f:
            meta = json.load(f)
        self.cases = meta["cases"]
        self.big = meta.get("big", [])

    def input(self, case):
        x = self.arrays["in_" + case["input"]]
        if case["kind"] == "decimate" and case["sample_size"] < 16:
            x = (x >> (16 - case["sample_size"])).astype(np.int16)
        if case["kind"] == "interpolate":
            x = x[:case["n_in"]]
        return x

    def expected(self, case):
        return self.arrays[case["key"]]


def headline():
    """tests/golden/headline_golden.json: whole-output SHA-256 digests of the benchmarked launches (reference decimator)"""
    with open(os.path.join(HERE, "headline_golden.json")) as f:
        return json.load(f)
