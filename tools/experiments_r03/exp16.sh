#!/bin/bash
cd $GRAFT_REPO_ROOT
ROUNDS=5 REPS=60 timeout 1200 bash tools/var_mfma.sh "-DMF_DMA=0" "-DMF_LOADER=0 -DMF_PAIR=0" "-DMF_LOADER=0" "-DMF_PAIR=0" "-" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp16_var.txt
cat gpurun_out/exp16_var.txt
