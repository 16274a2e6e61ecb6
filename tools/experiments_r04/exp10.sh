#!/bin/bash
cd $(dirname $0)/../..
L=tools/experiments_r04/lib
echo "== product"; PATHS=wave:0,wave:1024 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
for v in abl256 abl64; do echo "== $v"; SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so PATHS=wave:0,wave:1024 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate; done
