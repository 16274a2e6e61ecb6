# round 5, first GPU batch: parity of the new plumbing (overlap Rx, pipelined Tx, ring depth 3), then the A / B runs
set -x
O=gpurun_out/r05a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_headline.py tests/test_gpu_pipes.py tests/test_gpu_decim_mfma.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python tools/bench_ring.py 3 > $O/ring.log 2>&1
cat $O/ring.log
timeout 900 bash tools/sample_smi.sh $O/smi_rx_modes.txt python tools/bench_rx_modes.py > $O/rx_modes.log 2>&1
cat $O/rx_modes.log
timeout 900 bash tools/sample_smi.sh $O/smi_tx_modes.txt python tools/bench_tx_modes.py > $O/tx_modes.log 2>&1
cat $O/tx_modes.log
