#!/bin/bash
# evidence set of the round (R) (GPU box): bench line, kernel trace + PMC passes of the headline command and of the Tx pipe, kernel tables
cd $GRAFT_REPO_ROOT
R=r06
export LD_LIBRARY_PATH=$PWD/sdrdaemon_amd:$LD_LIBRARY_PATH
timeout 600 python bench.py --cpu-seconds 12 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver_cmd.json 2> gpurun_out/${R}_bench_driver_cmd.err
BENCH_ARGS="--no-configs --no-verify" bash tools/prof.sh > /dev/null 2>&1
cp gpurun_out/prof_bench/summary.txt gpurun_out/${R}_headline_rocprofv3_summary.txt
SDRHIP_DEC_MAX=32 bash tools/prof_cmd.sh tx python $PWD/tools/bench_kernels.py tx-random > /dev/null 2>&1
cp gpurun_out/prof_tx/summary.txt gpurun_out/${R}_tx_random_rocprofv3_summary.txt
python tools/bench_rx_modes.py > gpurun_out/${R}_rx_modes.txt 2>&1
python tools/bench_decim_paths.py > gpurun_out/${R}_decim_paths.txt 2>&1
python tools/bench_kernels.py decim interp fec > gpurun_out/${R}_kernels.txt 2>&1
PATHS=valu:0,wave:0 LS=1,2,3,4,5,6 REPS=30 python tools/bench_interp_paths.py 25 8 > gpurun_out/${R}_interp_paths.txt 2>&1
bash tools/bench_streams.sh > gpurun_out/${R}_streams.txt 2>&1
timeout 300 python tools/power_duty.py decim decim32 decim64 one27 interp interp2 > gpurun_out/${R}_power_duty.txt 2>&1
if [ -x tools/experiments_r06/bin/bench_dropin ]; then (timeout 60 tools/experiments_r06/bin/bench_dropin 65536; timeout 60 tools/experiments_r06/bin/bench_dropin 262144) > gpurun_out/${R}_dropin_cxx.txt 2>&1; fi
timeout 300 python tools/experiments_r06/tx_ab.py dec_plan kernel fused > gpurun_out/${R}_dec_plan_ab.txt 2>&1
python -c "
import json;d=json.loads([l for l in open('gpurun_out/${R}_bench.json') if l.startswith('{')][-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['roofline']['fec_encode_avg_launch_ms'],d['verified']['ok'],d['cpu_baseline']['value'],d.get('gpu_over_cpu_1core'),d.get('cpu_baseline_all_cores',{}).get('value'));[print(c['config'][:60],c['ms_per_step'],c['value'],c['roofline']['frac'],c['roofline']['avg_launch_ms'],c.get('decode_ms_per_step'),(c.get('verified') or {}).get('ok')) for c in d['configs']]"
grep "decim_mfma\|gf_encode\|frame_pack" gpurun_out/${R}_headline_rocprofv3_summary.txt | head -4
grep "interp_wave\|gf_decode" gpurun_out/${R}_tx_random_rocprofv3_summary.txt | head -4
