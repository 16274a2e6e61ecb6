# round 5, batch H: A / B on one box, five interleaved rounds: commit c30e36f (FFT encoder, K1m packs all four outputs in both lanes) against
# the tree (K1m: each lane of an I / Q pair finishes two outputs; rx_direct 0 / 1)
set -x
O=gpurun_out/r05h; mkdir -p $O
L=tools/experiments_r05/lib
for r in 1 2 3 4 5; do
  for v in c30 new0 new1; do
    if [ $v = c30 ]; then export SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_c30.so; unset SDRHIP_RX_DIRECT; else unset SDRHIP_LIB_PATH; export SDRHIP_RX_DIRECT=${v#new}; fi
    echo "== $v round $r" >> $O/rx.log
    MODES=immediate ROUNDS=1 timeout 300 python tools/bench_rx_modes.py 2>&1 | grep -i "immediate" >> $O/rx.log
  done
done
unset SDRHIP_RX_DIRECT SDRHIP_LIB_PATH
cat $O/rx.log
