#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_decim_mfma.py tests/test_gpu_headline.py -x -q 2>&1 | tail -5
ROUNDS=3 REPS=60 bash tools/var_mfma.sh "-" "-DMF_DMA=0" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp5_var.txt
cat gpurun_out/exp5_var.txt
