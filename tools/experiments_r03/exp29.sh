#!/bin/bash
cd $GRAFT_REPO_ROOT
make -s -C sdrdaemon_amd/csrc clean; make -s -C sdrdaemon_amd/csrc -j8 WITH_K5M=1 > /tmp/k5m.log 2>&1 || tail -5 /tmp/k5m.log
timeout 600 python -m pytest tests/test_gpu_interp_mfma.py tests/test_gpu_pipes.py -q 2>&1 | tail -3
