#!/bin/bash
# GPU box: interleaved span sweep of the matrix-core decimator: tools/sweep_span.sh <log2decim> "<span> <span> ..." (0 = planner)
L=${1:-4}
SPANS=${2:-"0 16896 22528 33792 45056"}
for r in 1 2 3; do for s in $SPANS; do echo "$s $(REPS=60 python tools/bench_decim_paths.py mfma:$s:$L 2>&1 | tail -1 | sed 's/.*span *[0-9]*: *//' | awk '{print $1}')"; done; done | sort -n | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k":"a[k]}' | sort -n
