#!/bin/bash
# batch 28 (timing only, wrong frames): the encoder inside the Rx pipe without its frame-copy stores (the fused framing)
cd /root/repo
for v in ebase nocopy ebase nocopy; do
  echo "== $v"
  SDRHIP_LIB_PATH=tools/experiments_r04/lib/libsdrhip_$v.so ROUNDS=1 python tools/bench_rx_modes.py 2>&1 | grep "immediate"
done
