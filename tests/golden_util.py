"""Loader for tests/golden/dsp_golden.{npz,json} (made by tests/golden/make_golden.py from the
compiled reference)."""
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self):
        self.arrays = np.load(os.path.join(HERE, "dsp_golden.npz"))
        with open(os.path.join(HERE, "dsp_golden.json")) as f:
            meta = json.load(f)
        self.cases = meta["cases"]
        self.big = meta["big"]

    def input(self, case):
        x = self.arrays["in_" + case["input"]]
        if case["kind"] == "decimate" and case["sample_size"] < 16:
            x = (x >> (16 - case["sample_size"])).astype(np.int16)
        if case["kind"] == "interpolate":
            x = x[:case["n_in"]]
        return x

    def expected(self, case):
        return self.arrays[case["key"]]
