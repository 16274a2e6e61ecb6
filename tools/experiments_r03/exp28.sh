#!/bin/bash
cd $GRAFT_REPO_ROOT
ROUNDS=3 REPS=60 timeout 1200 bash tools/var_mfma.sh "-" "-DMF_ABL=4096" "-DMF_LOADER=1 -DMF_ABL=4096" "-DMF_LOADER=2 -DMF_ABL=4096" "-DMF_LOADER=4 -DMF_ABL=4096" "-DMF_LOADER=2 -DMF_ABL=128" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp28_var.txt; cat gpurun_out/exp28_var.txt
