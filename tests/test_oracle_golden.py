"""Oracle vs the committed golden vectors (reference outputs): runs everywhere, incl. the GPU box
where neither /root/reference nor a compiler for it exists."""
import hashlib

import numpy as np
import pytest

import signals
from golden_util import Golden

G = Golden()


@pytest.mark.parametrize("flavour", ["eo1", "db"])
def test_oracle_matches_golden(oracle, flavour):
    n = 0
    for case in G.cases:
        if case["flavour"] != flavour:
            continue
        x = G.input(case)
        pos, outs = 0, []
        if case["kind"] == "decimate":
            d = oracle.decimators(case["bias"])
            ss_out = None
            for c in case["chunks"]:
                o, ss_out = d.decimate(case["log2"], case["fcpos"], case["sample_size"], x[pos:pos + c])
                outs.append(o)
                pos += c
            assert ss_out == case["sample_size_out"], case["key"]
        else:
            u = oracle.interpolators()
            for c in case["chunks"]:
                outs.append(u.interpolate(case["log2"], x[pos:pos + c]))
                pos += c
        assert np.array_equal(np.concatenate(outs), G.expected(case)), case["key"]
        n += 1
    assert n == 168


def test_oracle_matches_big_digests(oracle):
    for b in G.big:
        x = signals.noise(1 << 20, b["seed"])
        if b["kind"] == "decimate16_cen_blocks":
            d = oracle.decimators(b["bias"])
            h = hashlib.sha256()
            for i in range(16):
                o, _ = d.decimate(4, 2, 16, x[i * 65536:(i + 1) * 65536])
                h.update(o.tobytes())
            assert h.hexdigest() == b["sha256"]
        else:
            y = oracle.interpolators().interpolate(4, x[:b["n"]])
            assert hashlib.sha256(y.tobytes()).hexdigest() == b["sha256"]
