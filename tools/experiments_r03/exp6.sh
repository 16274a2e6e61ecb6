#!/bin/bash
cd $GRAFT_REPO_ROOT
ROUNDS=2 REPS=60 bash tools/var_mfma.sh "-" "-DMF_ABL=512" "-DMF_ABL=27" "-DMF_ABL=128" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp6_var.txt
cat gpurun_out/exp6_var.txt
bash tools/prof_mfma.sh mfma:0:4 > /dev/null 2>&1
grep -A40 "pass pmc1" gpurun_out/prof_mfma/summary.txt | grep "avg/dispatch\|vgpr"
