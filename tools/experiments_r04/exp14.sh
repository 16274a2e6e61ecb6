#!/bin/bash
# K4s: decode kernel timings (tx-random: 1024 frames, distinct random 24-erasure patterns) + counters
cd $(dirname $0)/../..
python -m pytest tests/test_gpu_fec.py tests/test_gpu_headline.py -x -q -k "fec or tx or dec" 2>&1 | tail -3
SDRHIP_DEC_MAX=32 python tools/bench_kernels.py tx-random 2>&1 | tail -4
SDRHIP_DEC_MAX=32 tools/prof_cmd.sh tx python $PWD/tools/bench_kernels.py tx-random > /dev/null 2>&1
grep -A14 "#### pass trace" gpurun_out/prof_tx/summary.txt | head -20
grep -B1 -A10 "PMC gf_decode128" gpurun_out/prof_tx/summary.txt | head -60
