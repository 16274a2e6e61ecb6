#!/bin/bash
# round-3 experiment batch 1: prefetch depth, stores, span sweep of K1m (decimate16_cen, 8 x 2^25)
cd $GRAFT_REPO_ROOT
ROUNDS=3 REPS=60 bash tools/var_mfma.sh "-" "-DMF_DEPTH=16" "-DMF_DEPTH=4" "-DMF_ABL=128" "-DMF_ABL=27" "-DMF_ABL=27 -DMF_DEPTH=16" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp1_var.txt
bash tools/sweep_span.sh 4 "0 16896 11264 8448" > gpurun_out/exp1_span.txt 2>&1
cat gpurun_out/exp1_var.txt gpurun_out/exp1_span.txt
