# round 5, batch M: block half as a template parameter (first half skips its 63 zero-constant multiplications) against the commit before
set -x
O=gpurun_out/r05m; mkdir -p $O
L=tools/experiments_r05/lib
timeout 900 python -m pytest tests/test_gpu_fec.py tests/test_gpu_headline.py tests/test_gpu_fuzz_slice.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do
  for v in head new; do
    if [ $v = new ]; then unset SDRHIP_LIB_PATH; else export SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so; fi
    echo "== $v round $r" >> $O/m.log
    timeout 300 python tools/bench_kernels.py fec 2>&1 | grep "fec_encode\|fec_decode" >> $O/m.log
    MODES=immediate ROUNDS=1 timeout 300 python tools/bench_rx_modes.py 2>&1 | grep -i "immediate" >> $O/m.log
    ROUNDS=1 timeout 300 python tools/bench_tx_modes.py 2>&1 | grep -i "immediate" >> $O/m.log
  done
done
unset SDRHIP_LIB_PATH
cat $O/m.log
