#!/bin/bash
# K5w with the straight-line whole-block loop (vmcnt(8) instead of a store drain per block)
cd $(dirname $0)/../..
L=tools/experiments_r04/lib
python -m pytest tests/test_gpu_interp_wave.py -x -q 2>&1 | tail -3
echo "== product"; PATHS=valu:0,wave:0,wave:1024 LS=4,5,3,2,6 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
for v in abl1 abl6 abl7; do echo "== $v"; SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so PATHS=wave:0 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate; done
echo "== counters wave"; tools/prof_interp.sh wave:0:4 2>&1 | tail -40
