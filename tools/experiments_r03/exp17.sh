#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_headline.py tests/test_gpu_pipes.py tests/test_gpu_decim_mfma.py -x -q 2>&1 | tail -8
