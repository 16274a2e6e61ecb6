#!/usr/bin/env python3
"""stagger with phase = workgroup mod 4 / 5 (in case consecutive workgroups share a CU) -- see stagger_sweep.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for mod in (4, 5, 8):
    env = dict(os.environ, SDRHIP_FEC_STAGGER_MOD=str(mod))
    print("==== fec_stagger_mod = %d" % mod, flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "experiments_r06", "stagger_sweep.py"), "1", "0", "4", "8", "16"], env=env)
