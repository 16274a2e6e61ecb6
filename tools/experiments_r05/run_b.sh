# round 5, batch B: ring-3 DMA issue in bursts (variant libraries), the Rx overlap mode on each; K5w with / without the Q-plane swizzle
set -x
O=gpurun_out/r05b; mkdir -p $O
L=tools/experiments_r05/lib
timeout 900 python -m pytest tests/test_gpu_interp_wave.py tests/test_gpu_pipes.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for v in product burst2 burst4 burst8; do
  if [ $v = product ]; then unset SDRHIP_LIB_PATH; else export SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so; fi
  echo "== $v" >> $O/ring.log
  NOSWEEP=1 timeout 300 python tools/bench_ring.py 2 >> $O/ring.log 2>&1
  echo "== $v" >> $O/rx_modes.log
  MODES=immediate,overlap ROUNDS=2 timeout 300 python tools/bench_rx_modes.py 2>&1 | grep -v "^   " >> $O/rx_modes.log
done
unset SDRHIP_LIB_PATH
cat $O/ring.log $O/rx_modes.log
for r in 1 2; do
  for v in product noswz; do
    if [ $v = product ]; then unset SDRHIP_LIB_PATH; else export SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so; fi
    echo "== $v (round $r)" >> $O/interp.log
    PATHS=wave:0 LS=4,3,2,5 timeout 300 python tools/bench_interp_paths.py >> $O/interp.log 2>&1
  done
done
unset SDRHIP_LIB_PATH
cat $O/interp.log
bash tools/prof_cmd.sh r05b_k5w python $PWD/tools/bench_interp_paths.py wave:0:4 > /dev/null 2>&1
cp gpurun_out/prof_r05b_k5w/summary.txt $O/k5w_swz_prof.txt
SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_noswz.so bash tools/prof_cmd.sh r05b_k5w0 python $PWD/tools/bench_interp_paths.py wave:0:4 > /dev/null 2>&1
cp gpurun_out/prof_r05b_k5w0/summary.txt $O/k5w_noswz_prof.txt
grep -A3 "LDS_BANK\|pass pmc2" $O/k5w_swz_prof.txt | head -40
