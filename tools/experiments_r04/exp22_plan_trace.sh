export TMPDIR=/tmp
ROOT=$PWD
python -m pytest tests/test_gpu_fec.py tests/test_gpu_headline.py -x -q 2>&1 | tail -2
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pt -o run -- python $ROOT/tools/bench_kernels.py tx-random > /tmp/pt.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) 2>&1 | grep -i "gf_dec\|interp_wave" | cut -c1-200
tail -3 /tmp/pt.log
