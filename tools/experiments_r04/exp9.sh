#!/bin/bash
cd $(dirname $0)/../..
echo "== plain"; PATHS=valu:0,wave:0,wave:1024 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
echo "== input freshly written by a copy kernel in front of every call"; PRECOPY=1 PATHS=valu:0,wave:0,wave:1024 LS=4 REPS=20 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate
