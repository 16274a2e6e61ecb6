#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_decim_mfma.py tests/test_gpu_headline.py -x -q 2>&1 | tail -3
ROUNDS=3 REPS=60 bash tools/var_mfma.sh "-" "-DMF_DMA=0" "-DMF_ABL=512" "-DMF_ABL=27" "-DMF_ABL=128" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp7_var.txt
cat gpurun_out/exp7_var.txt
