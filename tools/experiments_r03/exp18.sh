#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROOT=$PWD
cat > /tmp/fusedonly.py <<'PY'
import os, sys, time
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrdaemon_amd as sd
import signals
ctx = sd.Context(0)
S, n = 8, 1 << 25
x = torch.stack([signals.hash_noise_torch(n, 1000 + s, "cuda") for s in range(S)])
rx = sd.RxPipe(ctx, S, log2decim=4, nb_fec=32, pipelined=True)
for i in range(30):
    rx.process_view(x, i, 0)
torch.cuda.synchronize()
PY
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pf -o run -- python /tmp/fusedonly.py > /tmp/pf.log 2>&1
ROCPD_ALL_KERNELS=1 python $ROOT/tools/rocpd_summary.py $(find /tmp/pf -name "*.db" | head -1) 2>&1 | grep -A12 "rx_fused\|^kernel" | head -40
