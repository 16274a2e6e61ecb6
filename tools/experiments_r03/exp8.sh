#!/bin/bash
cd $GRAFT_REPO_ROOT
ROUNDS=2 REPS=60 bash tools/var_mfma.sh "-" "-DMF_NT=0" "-DMF_ABL=27" "-DMF_ABL=155" "-DMF_ABL=1179" "-DMF_ABL=128" "-DMF_DMA=0" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp8_var.txt
cat gpurun_out/exp8_var.txt
