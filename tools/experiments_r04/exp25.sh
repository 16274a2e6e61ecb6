#!/bin/bash
# batch 25: K5w with the last 1 / D of every stream in short segments of B blocks, dispatched last (tDsB; t0 = equal segments)
cd /root/repo
for v in ${VARIANTS:-t0 t4s4 t4s2 t4s8 t8s4 t3s4 t2s4 t0 t4s4 t4s2 t4s8 t8s4 t3s4 t2s4}; do
  echo "== $v"
  SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_$v.so PATHS=wave:0 LS=${LS:-4,3,5} python tools/bench_interp_paths.py 2>&1 | grep interpolate | cut -c1-100
done
