#!/bin/bash
cd $GRAFT_REPO_ROOT
BENCH_ARGS="--no-configs --no-verify" bash tools/prof.sh > /dev/null 2>&1
cp gpurun_out/prof_bench/summary.txt gpurun_out/r03_headline_rocprofv3_summary.txt
bash tools/prof_cmd.sh tx python $PWD/tools/bench_kernels.py tx-random > /dev/null 2>&1
cp gpurun_out/prof_tx/summary.txt gpurun_out/r03_tx_random_rocprofv3_summary.txt
python bench.py --cpu-seconds 12 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python tools/bench_decim_paths.py > gpurun_out/r03_decim_paths.txt 2>&1
python tools/bench_kernels.py decim interp > gpurun_out/r03_kernels.txt 2>&1
python tools/bench_rx_modes.py > gpurun_out/r03_rx_modes.txt 2>&1
head -c 600 gpurun_out/r03_bench.json; echo; grep "decim_mfma\|gf_encode\|frame_pack" gpurun_out/r03_headline_rocprofv3_summary.txt | head -8
