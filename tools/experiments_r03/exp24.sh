#!/bin/bash
cd $GRAFT_REPO_ROOT
CFG="mfma:0:4 mfma:0:5 mfma:0:6" ROUNDS=3 REPS=60 timeout 1200 bash tools/var_mfma.sh "-" "-DMF_DMA=0" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp24_var.txt; cat gpurun_out/exp24_var.txt
