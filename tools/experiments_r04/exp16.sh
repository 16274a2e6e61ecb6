#!/bin/bash
# K1m's "two box states" (VERDICT r3 #6b): three 3000-step runs of the headline step with rocm-smi sampled at ~10 Hz beside them
cd $(dirname $0)/../..
for r in 1 2 3; do
  tools/sample_smi.sh gpurun_out/r04_smi_run$r.txt python bench.py --steps 6000 --warmup 20 --no-configs --cpu-seconds 0 --no-verify > gpurun_out/r04_smi_bench$r.json 2> /dev/null
  python - <<PY
import json
d = json.loads([ln for ln in open("gpurun_out/r04_smi_bench$r.json").read().splitlines() if ln.startswith("{")][-1])
print("run $r: ms_per_step", d["ms_per_step"], "K1m avg launch ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"])
PY
  tail -n +2 gpurun_out/r04_smi_run$r.txt | awk '$2 > 600 {for(i=2;i<=NF;i++){s[i]+=$i}; n++} END {printf "   mean over the loaded part:"; for(i=2;i<=NF;i++) printf " %.1f", s[i]/n; print ""}'
  head -1 gpurun_out/r04_smi_run$r.txt
done
