#!/bin/bash
# batch 20: waves per workgroup for K5w (independent segments; co-started waves => their load clusters coincide in time)
cd /root/repo
for v in wpw1 wpw4 wpw8 wpw1 wpw4 wpw8; do
  echo "== $v"
  SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_$v.so PATHS=wave:0 LS=4,3,5 python tools/bench_interp_paths.py 2>&1 | grep -v "^$"
done
