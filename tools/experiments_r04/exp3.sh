#!/bin/bash
# does the write stream of K5w suffer from the NUMBER of concurrent streams?  (a) store-only probe, (b) K5w with fewer waves per CU
cd $(dirname $0)/../..
L=tools/experiments_r04/lib
$L/store_streams
for pad in 0 2560 5632 8192 12288; do
  echo "== product pad $pad"; PAD=$pad PATHS=wave:0,wave:1024 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
  echo "== abl6 pad $pad"; PAD=$pad SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_abl6.so PATHS=wave:0 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
done
