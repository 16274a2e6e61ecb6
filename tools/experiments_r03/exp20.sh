#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for p in syndrome dense; do for m in 128 32; do echo "== dec_path $p  dec_max_rows $m"; SDRHIP_DEC_PATH=$p SDRHIP_DEC_MAX=$m python tools/bench_kernels.py tx-random 2>&1 | grep "decode\|tx pipe, a random"; done; done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ptx -o run -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py tx-random > /tmp/ptx.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/ptx -name "*.db" | head -1) 2>&1 | head -12
