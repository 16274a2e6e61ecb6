cd $GRAFT_REPO_ROOT
BENCH_ARGS="--no-verify" bash tools/prof.sh > /dev/null 2>&1
cp gpurun_out/prof_bench/summary.txt gpurun_out/r04_bench_configs_rocprofv3_summary.txt
grep -n "rx_fused\|decim_mfma\|gf_encode128_pack\|interp_wave\|gf_decode128" gpurun_out/r04_bench_configs_rocprofv3_summary.txt | head -40
