#!/bin/bash
# A / B of the non-temporal loads of the FFT decoder / encoder (variant libraries: tools/experiments_r05/build_variant.sh <name> -D...), interleaved
L=tools/experiments_r05/lib
for round in 1 2; do
  python tools/experiments_r06/lib_ab.py "product (decoder loads nt)"
  for v in "$@"; do SDRHIP_LIB_PATH=$L/libsdrhip_$v.so python tools/experiments_r06/lib_ab.py "variant $v"; done
done 2>&1 | grep -v amdgpu.ids
