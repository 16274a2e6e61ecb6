#!/bin/bash
cd $(dirname $0)/../..
L=tools/experiments_r04/lib
for r in 1 2; do
echo "== product"; PATHS=wave:0 LS=4,3,5 REPS=30 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
echo "== wpe5"; SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_wpe5.so PATHS=wave:0 LS=4,3,5 REPS=30 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
done
