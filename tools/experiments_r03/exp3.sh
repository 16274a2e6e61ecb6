#!/bin/bash
cd $GRAFT_REPO_ROOT
ROUNDS=3 REPS=60 bash tools/var_mfma.sh "-" "-DMF_ABL=512" "-DMF_ABL=640" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp3_var.txt
cat gpurun_out/exp3_var.txt
