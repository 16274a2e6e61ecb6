#!/bin/bash
cd $GRAFT_REPO_ROOT
ROUNDS=3 REPS=60 bash tools/var_mfma.sh "-" "-DMF_ABL=256" "-DMF_ABL=384" "-DMF_ABL=128" "-DMF_ABL=283" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp2_var.txt
cat gpurun_out/exp2_var.txt
