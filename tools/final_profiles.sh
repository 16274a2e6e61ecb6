#!/bin/bash
# evidence set of the round (R) (GPU box): bench line, kernel trace + PMC passes of the headline command and of the Tx pipe, kernel tables
cd $GRAFT_REPO_ROOT
R=r05
python bench.py --cpu-seconds 12 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
BENCH_ARGS="--no-configs --no-verify" bash tools/prof.sh > /dev/null 2>&1
cp gpurun_out/prof_bench/summary.txt gpurun_out/${R}_headline_rocprofv3_summary.txt
SDRHIP_DEC_MAX=32 bash tools/prof_cmd.sh tx python $PWD/tools/bench_kernels.py tx-random > /dev/null 2>&1
cp gpurun_out/prof_tx/summary.txt gpurun_out/${R}_tx_random_rocprofv3_summary.txt
python tools/bench_rx_modes.py > gpurun_out/${R}_rx_modes.txt 2>&1
python tools/bench_decim_paths.py > gpurun_out/${R}_decim_paths.txt 2>&1
python tools/bench_kernels.py decim interp fec > gpurun_out/${R}_kernels.txt 2>&1
PATHS=valu:0,wave:0 LS=1,2,3,4,5,6 REPS=30 python tools/bench_interp_paths.py 25 8 > gpurun_out/${R}_interp_paths.txt 2>&1
bash tools/bench_streams.sh > gpurun_out/${R}_streams.txt 2>&1
python -c "
import json;d=json.loads([l for l in open('gpurun_out/${R}_bench.json') if l.startswith('{')][-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['roofline']['fec_encode_avg_launch_ms'],d['verified']['ok'],d['cpu_baseline']['value'],d.get('gpu_over_cpu_1core'),d.get('cpu_baseline_all_cores',{}).get('value'));[print(c['config'][:60],c['ms_per_step'],c['value'],c['roofline']['frac'],c['roofline']['avg_launch_ms'],c.get('decode_ms_per_step'),(c.get('verified') or {}).get('ok')) for c in d['configs']]"
grep "decim_mfma\|gf_encode\|frame_pack" gpurun_out/${R}_headline_rocprofv3_summary.txt | head -4
grep "interp_wave\|gf_decode" gpurun_out/${R}_tx_random_rocprofv3_summary.txt | head -4
