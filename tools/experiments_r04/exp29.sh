#!/bin/bash
# batch 29: K1m on fewer CUs (longer spans): does a power-limited launch gain from idle CUs?
cd /root/repo
for sp in 0 34816 36864 38912 40960 45056 0; do
  python tools/bench_decim_paths.py mfma:$sp:4 2>&1 | grep decimate | cut -c1-100
done
