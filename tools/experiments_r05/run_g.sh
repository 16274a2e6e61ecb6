# round 5, batch G: the matrix-core decimator storing straight into the frame layout (rx_direct = 1) against stream order + K2 + fused copy
set -x
O=gpurun_out/r05g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_decim_mfma.py tests/test_gpu_decimators.py tests/test_gpu_pipes.py tests/test_gpu_headline.py tests/test_gpu_fuzz_slice.py tests/test_gpu_udp_adapters.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for r in 1 2 3; do
  for v in 0 1; do
    export SDRHIP_RX_DIRECT=$v
    echo "== rx_direct $v round $r" >> $O/rx.log
    MODES=immediate ROUNDS=1 timeout 300 python tools/bench_rx_modes.py 2>&1 | grep -i "immediate" >> $O/rx.log
  done
done
unset SDRHIP_RX_DIRECT
cat $O/rx.log
