# round 5, batch F: FFT encoder, workgroups per CU limited by LDS (4 = product, 3, 2): do rounds of workgroups overlap the load / compute phases?
set -x
O=gpurun_out/r05f; mkdir -p $O
L=tools/experiments_r05/lib
for r in 1 2 3; do
  for v in product wg3 wg2; do
    if [ $v = product ]; then unset SDRHIP_LIB_PATH; else export SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so; fi
    echo "== $v round $r" >> $O/enc.log
    timeout 300 python tools/bench_kernels.py fec 2>&1 | grep fec_encode >> $O/enc.log
    MODES=immediate ROUNDS=1 timeout 300 python tools/bench_rx_modes.py 2>&1 | grep -i "immediate" >> $O/enc.log
  done
done
unset SDRHIP_LIB_PATH
cat $O/enc.log
