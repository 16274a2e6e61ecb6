#!/bin/bash
cd $GRAFT_REPO_ROOT
ROUNDS=2 REPS=60 timeout 900 bash tools/var_mfma.sh "-DMF_LOADER=0 -DMF_ABL=27" "-DMF_LOADER=0 -DMF_ABL=27 -DMF_STV=1" "-DMF_LOADER=0 -DMF_ABL=27 -DMF_STV=2" "-DMF_LOADER=0 -DMF_ABL=27 -DMF_STV=3" "-DMF_LOADER=0 -DMF_ABL=155" "-DMF_ABL=27" "-DMF_ABL=155" "-DMF_LOADER=0 -DMF_STV=1" "-DMF_LOADER=0" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp12_var.txt
cat gpurun_out/exp12_var.txt
