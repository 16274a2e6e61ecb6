#!/bin/bash
# batch 24: K5w start of a wave: warm-up samples + bank state loaded with the segment's cluster (one round trip instead of three)
cd /root/repo
for v in base early base early base early; do
  echo "== $v"
  SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_$v.so PATHS=wave:0 LS=4,3,5,2 python tools/bench_interp_paths.py 2>&1 | grep interpolate
done
