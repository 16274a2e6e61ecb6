#!/bin/bash
# GF kernels with the leaf tables fetched one leaf ahead: parity + timings
cd $(dirname $0)/../..
python -m pytest tests/test_gpu_fec.py tests/test_gpu_headline.py tests/test_gpu_pipes.py -x -q 2>&1 | tail -3
python tools/bench_kernels.py fec 2>&1 | tail -2
SDRHIP_DEC_MAX=32 python tools/bench_kernels.py tx-random 2>&1 | tail -3
python tools/bench_rx_modes.py 2>&1 | tail -8
