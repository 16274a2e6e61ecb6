#!/bin/bash
# batch 26: K1m's VALU pieces (stream heads and tails) on as many workgroups as the planner left CUs free, several pieces each
cd /root/repo
SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_mfstamps.so python tools/experiments_r04/k1m_stamps.py 4 2>&1 | tail -36 | grep -v "^  xcc\|slowest"
for v in pbase pnew pbase pnew pbase pnew; do
  echo "== $v"
  SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_$v.so python tools/bench_decim_paths.py mfma:0:4 2>&1 | grep decimate
done
