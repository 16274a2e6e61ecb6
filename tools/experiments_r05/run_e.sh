# round 5, batch E: the additive-FFT encoder (gf_encode128_fft.h) against the Karatsuba walk: parity tests, kernel time, Rx step
set -x
O=gpurun_out/r05e; mkdir -p $O
L=tools/experiments_r05/lib
timeout 1200 python -m pytest tests/test_gpu_fec.py tests/test_gpu_pipes.py tests/test_gpu_headline.py tests/test_gpu_fuzz_slice.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for r in 1 2; do
  for v in karatsuba fft; do
    if [ $v = fftskip ]; then export SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_fftskip.so SDRHIP_ENC_PATH=fft; else unset SDRHIP_LIB_PATH; export SDRHIP_ENC_PATH=$v; fi
    echo "== $v round $r" >> $O/enc.log
    timeout 300 python tools/bench_kernels.py fec 2>&1 | grep fec_encode >> $O/enc.log
    MODES=immediate ROUNDS=1 timeout 300 python tools/bench_rx_modes.py 2>&1 | grep -i "immediate" >> $O/enc.log
  done
done
unset SDRHIP_LIB_PATH SDRHIP_ENC_PATH
cat $O/enc.log
