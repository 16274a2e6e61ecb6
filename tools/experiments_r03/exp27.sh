#!/bin/bash
cd $GRAFT_REPO_ROOT
rm -f sdrdaemon_amd/csrc/build/decim_mfma.hip.o; make -s -C sdrdaemon_amd/csrc EXTRA="-DMF_LOADER=2" > /dev/null 2>&1
timeout 300 python -m pytest tests/test_gpu_decim_mfma.py tests/test_gpu_headline.py -x -q 2>&1 | tail -3
ROUNDS=3 REPS=60 timeout 900 bash tools/var_mfma.sh "-" "-DMF_LOADER=2" "-DMF_LOADER=4" "-DMF_LOADER=1" > /dev/null 2>&1
cp gpurun_out/var_mfma.txt gpurun_out/exp27_var.txt; cat gpurun_out/exp27_var.txt
