#!/bin/bash
# usage (GPU box): tools/sample_smi.sh <out.txt> <command...>  -- runs the command while sampling rocm-smi (clocks, power, temperatures)
# at ~10 Hz: one line per sample "t_ms sclk_MHz mclk_MHz fclk_MHz socket_W edge_C junction_C mem_C"
OUT=$1; shift
( t0=$(date +%s%N)
  while :; do
    s=$(rocm-smi -d 0 --showclocks --showpower --showtemp --json 2>/dev/null)
    t=$(( ($(date +%s%N) - t0) / 1000000 ))
    echo "$t $s" >> $OUT.raw
    sleep 0.05
  done ) &
SP=$!
"$@"
RC=$?
kill $SP 2>/dev/null
python3 - "$OUT" <<'PY'
import json, re, sys
out = sys.argv[1]
rows = []
for ln in open(out + ".raw"):
    t, _, js = ln.partition(" ")
    try:
        d = json.loads(js)["card0"]
    except Exception:
        continue
    def num(k):
        v = d.get(k, "")
        m = re.search(r"([0-9.]+)", str(v))
        return float(m.group(1)) if m else float("nan")
    keys = {k: num(k) for k in d}
    rows.append((int(t), keys))
names = sorted({k for _, ks in rows for k in ks if re.search(r"sclk|mclk|fclk|socclk|Power|Temperature", k)})
with open(out, "w") as f:
    f.write("# t_ms " + " | ".join(names) + "\n")
    for t, ks in rows:
        f.write("%6d " % t + " ".join("%8.1f" % ks.get(n, float("nan")) for n in names) + "\n")
print("smi samples:", len(rows), "->", out)
PY
rm -f $OUT.raw
exit $RC
