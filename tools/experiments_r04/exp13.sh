#!/bin/bash
cd $(dirname $0)/../..
L=tools/experiments_r04/lib
echo "== product (8 pairs)"; PATHS=valu:0,wave:0 LS=4 REPS=30 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
for v in p7 p6 p4; do echo "== $v"; SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so PATHS=wave:0 LS=4,5,3 REPS=30 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate; done
echo "== product again"; PATHS=wave:0 LS=4,5,3 REPS=30 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
