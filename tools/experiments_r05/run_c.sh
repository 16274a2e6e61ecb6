# round 5, batch C: K1m's LDS-DMA issue in bursts, ring depth 4 (the product geometry): 4 interleaved rounds of the variant libraries,
# kernel alone and the immediate Rx step; then bench.py's new JSON (box state, verified lines) as a smoke run
set -x
O=gpurun_out/r05c; mkdir -p $O
L=tools/experiments_r05/lib
for r in 1 2 3 4; do
  for v in product burst2 burst4 burst8; do
    if [ $v = product ]; then unset SDRHIP_LIB_PATH; else export SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so; fi
    echo "== $v round $r" >> $O/ring.log
    RINGS=4 NOSWEEP=1 timeout 300 python tools/bench_ring.py 1 2>&1 | grep "ring depth" >> $O/ring.log
    echo "== $v round $r" >> $O/rx_modes.log
    MODES=immediate ROUNDS=1 timeout 300 python tools/bench_rx_modes.py 2>&1 | grep -v "amdgpu" >> $O/rx_modes.log
  done
done
unset SDRHIP_LIB_PATH
cat $O/ring.log $O/rx_modes.log
timeout 900 python bench.py --cpu-seconds 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05c/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "verified", d["verified"]["ok"], "box", d.get("box"))
for c in d.get("configs", []):
    print(" -", c["config"][:70], c["ms_per_step"], c["roofline"]["frac"], (c.get("verified") or {}).get("ok"))
PY
timeout 1500 python -m pytest tests/test_gpu_fuzz_slice.py tests/test_distributed.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
