#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_decim_mfma.py tests/test_gpu_headline.py -x -q 2>&1 | tail -4
ROUNDS=2 REPS=60 timeout 900 bash tools/var_mfma.sh "-DMF_LOADER=0 -DMF_ABL=27" "-DMF_LOADER=0 -DMF_ABL=27 -DMF_PAIR=0" "-DMF_LOADER=0" "-DMF_LOADER=0 -DMF_PAIR=0" "-" "-DMF_DMA=0" "-DMF_DMA=0 -DMF_PAIR=0" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp15_var.txt
cat gpurun_out/exp15_var.txt
