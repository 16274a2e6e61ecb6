#!/bin/bash
# GPU box: headline pipe for several bank sizes and span overrides: tools/bench_streams.sh "8 16" "0 31744 32768 33792"
for ns in ${1:-8 16 24 32 64}; do for sp in ${2:-0}; do
  if [ "$sp" = "0" ]; then unset SDRHIP_MFMA_SPAN; else export SDRHIP_MFMA_SPAN=$sp; fi
  python bench.py --cpu-seconds 0 --streams $ns --no-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($ns, 'streams span $sp:', d['value'], d['ms_per_step'], 'K1m per 2^28 samples: %.4f ms' % (d['roofline']['avg_launch_ms'] * 8 / $ns))"
done; done
