# round 5, batch D: the whole -m gpu suite on the current tree; the asynchronous Tx entry against the synchronous calls;
# decimate32 / 64 on the LDS-DMA ring (variant library); bench.py
set -x
O=gpurun_out/r05d; mkdir -p $O
L=tools/experiments_r05/lib
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python tools/bench_host_block.py tx > $O/host_block_tx.log 2>&1
cat $O/host_block_tx.log
for r in 1 2; do
  for v in product dma56b8; do
    if [ $v = product ]; then unset SDRHIP_LIB_PATH; else export SDRHIP_LIB_PATH=$PWD/$L/libsdrhip_$v.so; fi
    for Lg in 5 6; do
      echo "== $v round $r" >> $O/decim56.log
      timeout 300 python tools/bench_decim_paths.py mfma:0:$Lg 2>&1 | grep decimate >> $O/decim56.log
    done
  done
done
unset SDRHIP_LIB_PATH
cat $O/decim56.log
timeout 900 python bench.py --cpu-seconds 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05d/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "verified", d["verified"]["ok"], "box", d.get("box"))
for c in d.get("configs", []):
    print(" -", c["config"][:70], c["ms_per_step"], c["roofline"]["frac"], (c.get("verified") or {}).get("ok"))
PY
