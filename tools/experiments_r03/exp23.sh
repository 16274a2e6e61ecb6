#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pipes.py tests/test_gpu_headline.py tests/test_gpu_decim_mfma.py tests/test_gpu_udp_adapters.py tests/test_gpu_testsource.py -x -q 2>&1 | tail -4
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/pb -o run -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --no-configs --no-verify > /tmp/pb.log 2>&1; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/pb -name '*.db' | head -1) 2>&1 | head -6 )
