# round 5, batch L: the syndrome decoder with the additive-FFT walk (enc_path fft) against the Karatsuba walk: parity tests, kernel time, Tx step
set -x
O=gpurun_out/r05l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fec.py tests/test_gpu_pipes.py tests/test_gpu_headline.py tests/test_gpu_fuzz_slice.py tests/test_gpu_udp_adapters.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for r in 1 2 3; do
  for v in karatsuba fft; do
    export SDRHIP_ENC_PATH=$v
    echo "== $v round $r" >> $O/dec.log
    timeout 300 python tools/bench_kernels.py fec 2>&1 | grep -i "fec_decode\|tx pipe\|of which" >> $O/dec.log
  done
done
unset SDRHIP_ENC_PATH
cat $O/dec.log
