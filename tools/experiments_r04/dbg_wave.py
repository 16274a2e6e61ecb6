import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import signals, sdrdaemon_amd as sd
from oracle_lib import Oracle
ctx = sd.Context(0); orc = Oracle()
ctx.set_option("interp_path", "wave")
x = signals.noise(4096, 5)
for L in (4, 5, 6):
    for n in (128, 256, 384, 512, 1024):
        a = sd.Interpolators(ctx, 1).interpolate(L, x[:n]); b = orc.interpolators().interpolate(L, x[:n])
        bad = np.argwhere((a != b).any(axis=1))[:, 0]
        print("L", L, "n", n, "mismatches", len(bad), "first", bad[:6], "last", bad[-3:], "blocks(out/2048..)", sorted(set((bad >> (L + 7)).tolist()))[:8])
