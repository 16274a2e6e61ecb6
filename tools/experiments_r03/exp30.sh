#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_decim_mfma.py tests/test_gpu_headline.py tests/test_gpu_pipes.py -x -q 2>&1 | tail -3
ROUNDS=4 REPS=60 timeout 900 bash tools/var_mfma.sh "-" "-DMF_NOCLAMP" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp30_var.txt; cat gpurun_out/exp30_var.txt
python tools/bench_decim_paths.py mfma:0:4 27 1 2>&1 | tail -1
