# round 5, batch J: FFT encoder at 5 (product) against 4 waves per SIMD, Rx step in the direct arrangement; bench.py
set -x
O=gpurun_out/r05j; mkdir -p $O
L=tools/experiments_r05/lib
for r in 1 2 3; do
  for v in product w4; do
    if [ $v = product ]; then unset SDRHIP_LIB_PATH; else export SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so; fi
    echo "== $v round $r" >> $O/rx.log
    MODES=immediate ROUNDS=1 timeout 300 python tools/bench_rx_modes.py 2>&1 | grep -i "immediate" >> $O/rx.log
  done
done
unset SDRHIP_LIB_PATH
cat $O/rx.log
timeout 900 python bench.py --cpu-seconds 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05j/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "verified", d["verified"]["ok"], "box", d.get("box"))
print(json.dumps(d["roofline"])[:1500])
for c in d.get("configs", []):
    print(" -", c["config"][:70], c["ms_per_step"], c["roofline"]["frac"], (c.get("verified") or {}).get("ok"))
PY
