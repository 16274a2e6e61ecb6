#!/bin/bash
cd $(dirname $0)/../..
python -m pytest tests/test_gpu_interp_wave.py tests/test_gpu_headline.py -x -q -k "wave or tx" 2>&1 | tail -2
for r in 1 2 3; do PATHS=wave:0 LS=4 REPS=30 python tools/bench_interp_paths.py 25 8 2>&1 | grep interpolate; done
python bench.py --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
c=d['configs'][-1]; print('tx', c['ms_per_step'], c['roofline']['avg_launch_ms'], c['roofline']['frac'], c['decode_ms_per_step'], c['verified']['ok'])"
