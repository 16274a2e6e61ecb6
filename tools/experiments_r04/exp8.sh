#!/bin/bash
cd $(dirname $0)/../..
L=tools/experiments_r04/lib
echo "== product (WG 4)"; PATHS=valu:0,wave:0,wave:1024 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
for v in g6 g8; do echo "== $v"; SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so PATHS=wave:0,wave:1536,wave:2048,wave:3072,wave:4096 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate; done
