#!/bin/bash
# round-3 evidence set (GPU box): bench line, kernel trace + PMC passes of the headline command and of the Tx pipe, kernel tables
cd $GRAFT_REPO_ROOT
python bench.py --cpu-seconds 12 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
BENCH_ARGS="--no-configs --no-verify" bash tools/prof.sh > /dev/null 2>&1
cp gpurun_out/prof_bench/summary.txt gpurun_out/r03_headline_rocprofv3_summary.txt
SDRHIP_DEC_MAX=32 bash tools/prof_cmd.sh tx python $PWD/tools/bench_kernels.py tx-random > /dev/null 2>&1
cp gpurun_out/prof_tx/summary.txt gpurun_out/r03_tx_random_rocprofv3_summary.txt
python tools/bench_rx_modes.py > gpurun_out/r03_rx_modes.txt 2>&1
python tools/bench_decim_paths.py > gpurun_out/r03_decim_paths.txt 2>&1
python tools/bench_kernels.py decim interp > gpurun_out/r03_kernels.txt 2>&1
python tools/bench_host_block.py > gpurun_out/r03_host_block.txt 2>&1
bash tools/bench_streams.sh > gpurun_out/r03_streams.txt 2>&1
python -c "
import json;d=json.load(open('gpurun_out/r03_bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['roofline']['fec_encode_avg_launch_ms'],d['verified']['ok'],d['cpu_baseline']['value'],d.get('gpu_over_cpu_1core'),d.get('cpu_baseline_all_cores',{}).get('value'));[print(c['config'][:60],c['ms_per_step'],c['value'],c['roofline']['frac'],c['roofline']['avg_launch_ms'],c.get('decode_ms_per_step')) for c in d['configs']]"
grep "decim_mfma\|gf_encode\|frame_pack" gpurun_out/r03_headline_rocprofv3_summary.txt | head -4
