#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_decim_mfma.py tests/test_gpu_headline.py -x -q 2>&1 | tail -3
ROUNDS=4 REPS=60 timeout 900 bash tools/var_mfma.sh "-" "-DMF_FRONT2=0" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp25_var.txt; cat gpurun_out/exp25_var.txt
