#!/bin/bash
# batch 27: the Karatsuba walk's 4-point nodes in the selector domain (encoder and decoder share the walk)
cd /root/repo
python -m pytest tests/test_gpu_fec.py tests/test_gpu_headline.py -x -q 2>&1 | tail -2
for v in ${VARIANTS:-kbase ksel4 kbase ksel4 kbase ksel4}; do
  echo "== $v"
  SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_$v.so python tools/power_duty.py enc 2>&1 | grep "gap      0" | head -1
  SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_$v.so SDRHIP_DEC_MAX=32 python tools/bench_kernels.py tx-random 2>&1 | grep "of which\|random 24" | head -2
done
