#!/bin/bash
cd $(dirname $0)/../..
L=tools/experiments_r04/lib
echo "== product (WG 4)"; PATHS=valu:0,wave:0,wave:512,wave:1024,wave:1536,wave:3072 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
for v in abl64 abl128; do echo "== $v"; SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so PATHS=wave:0,wave:1024 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate; done
for pad in 1024 3072; do echo "== product pad $pad"; PAD=$pad PATHS=wave:0,wave:1024 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate; done
