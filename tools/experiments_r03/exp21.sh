#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in 4 3; do
rm -f sdrdaemon_amd/csrc/build/gf_kernels.hip.o; make -s -C sdrdaemon_amd/csrc EXTRA="-DDEC128_WPE=$w" > /dev/null 2>&1
for r in 1 2; do
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ptx$w$r -o run -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py tx-random > /tmp/ptx.log 2>&1; echo "WPE=$w: $(python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/ptx$w$r -name '*.db' | head -1) 2>&1 | grep 'gf_decode128\|gf_decode_plan' | awk '{print $1, $(NF-3)}' | tr '\n' ' ')" )
done; done
