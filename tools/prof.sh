#!/bin/bash
# Runs on the GPU box (via gpurun): profile of the default bench command -> gpurun_out/prof_bench/summary.txt
exec bash tools/prof_cmd.sh bench python $PWD/bench.py --cpu-seconds 0 $BENCH_ARGS
