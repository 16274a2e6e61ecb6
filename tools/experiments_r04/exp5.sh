#!/bin/bash
# K5w: is it the input READ stream mixed into the write stream?  32 = cached loads, 64 = no loads; 38 / 70 = the same without FIR arithmetic
cd $(dirname $0)/../..
L=tools/experiments_r04/lib
echo "== product"; PATHS=valu:0,wave:0 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
for v in abl32 abl64 abl38 abl70 abl6; do echo "== $v"; SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so PATHS=wave:0 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate; done
