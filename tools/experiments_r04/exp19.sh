#!/bin/bash
cd $(dirname $0)/../..
L=tools/experiments_r04/lib
for r in 1 2; do
echo "== product"; PATHS=wave:0 LS=4,3,5 REPS=30 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
echo "== regs (56 VGPR + 32 AGPR: 5 waves per SIMD)"; SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_regs.so PATHS=wave:0 LS=4,3,5 REPS=30 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
done
SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_regs.so python -m pytest tests/test_gpu_interp_wave.py -x -q 2>&1 | tail -2
