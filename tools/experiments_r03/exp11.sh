#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_decim_mfma.py tests/test_gpu_headline.py -x -q 2>&1 | tail -5
ROUNDS=3 REPS=60 timeout 600 bash tools/var_mfma.sh "-" "-DMF_LOADER=0" "-DMF_DMA=0" "-DMF_ABL=128" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp11_var.txt
cat gpurun_out/exp11_var.txt
