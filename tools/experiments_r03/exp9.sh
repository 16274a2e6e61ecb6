#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/prof_mfma.sh mfma:0:4 > /dev/null 2>&1
grep "avg/dispatch\|vgpr\|decim_mfma_kernel" gpurun_out/prof_mfma/summary.txt | head -80
