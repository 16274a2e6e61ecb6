#!/bin/bash
# K5w first light: segment lengths (power-of-two strides or not), ablations (WRONG results, timing only), counters
cd $(dirname $0)/../..
L=tools/experiments_r04/lib
echo "== product"; PATHS=valu:0,wave:0,wave:1920,wave:2176,wave:1152,wave:896 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
for v in abl1 abl6 abl7 abl8 abl16; do echo "== $v"; SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so PATHS=wave:0 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate; done
echo "== counters wave"; tools/prof_interp.sh wave:0:4 2>&1 | tail -40
echo "== counters valu"; tools/prof_interp.sh valu:0:4 2>&1 | tail -40
