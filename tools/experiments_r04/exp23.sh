#!/bin/bash
# batch 23: K5w segments of 2 x 8 / 12 / 16 blocks (32 / 48 / 64 AGPRs) for the short cascades interpolate4 / 8
cd /root/repo
for v in ${VARIANTS:-wp8 wp12 wp16 wp8 wp12 wp16}; do
  echo "== $v"
  SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_$v.so PATHS=wave:0 LS=2,3 python tools/bench_interp_paths.py 2>&1 | grep interpolate
done
