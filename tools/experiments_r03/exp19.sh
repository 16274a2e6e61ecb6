#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 2 3; do
rm -f sdrdaemon_amd/csrc/build/decim_mfma.hip.o; make -s -C sdrdaemon_amd/csrc EXTRA="-DMF_FUSED_WPE=$w" > /dev/null 2>&1
echo "== MF_FUSED_WPE=$w"; ROUNDS=1 python tools/bench_rx_modes.py 2>&1 | grep "fused"
done
