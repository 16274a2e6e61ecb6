#!/bin/bash
cd $GRAFT_REPO_ROOT
ROUNDS=3 REPS=60 timeout 900 bash tools/var_mfma.sh "-DMF_ABL=27" "-DMF_ABL=4123" "-" "-DMF_ABL=4096" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp26_var.txt; cat gpurun_out/exp26_var.txt
