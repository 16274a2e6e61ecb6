#!/bin/bash
# GPU box: timing of build variants of the matrix-core decimator: tools/var_mfma.sh "<flags 1>" "<flags 2>" ...
# (each argument = the EXTRA flags of one build; "-" = none).  All variants are built first, then timed in
# ROUNDS (default 3) interleaved rounds (clock / thermal drift hits them alike); per variant: min and all samples.
#   -> gpurun_out/var_mfma.txt
ROOT=$PWD
OUT=$ROOT/gpurun_out/var_mfma.txt
CFG=${CFG:-mfma:0:4}
ROUNDS=${ROUNDS:-3}
export REPS=${REPS:-60}
mkdir -p $ROOT/gpurun_out /tmp/var; : > $OUT
i=0
for flags in "$@"; do
    [ "$flags" = "-" ] && flags=""
    rm -f sdrdaemon_amd/csrc/build/decim_mfma.hip.o
    if make -s -C sdrdaemon_amd/csrc EXTRA="$flags" > /tmp/var/build_$i.log 2>&1; then cp sdrdaemon_amd/libsdrhip.so /tmp/var/lib_$i.so; else echo "build failed: $flags" >> $OUT; fi
    i=$((i+1))
done
n=$i
for r in $(seq $ROUNDS); do
    for i in $(seq 0 $((n-1))); do
        [ -f /tmp/var/lib_$i.so ] || continue
        cp /tmp/var/lib_$i.so sdrdaemon_amd/libsdrhip.so
        for c in $CFG; do python tools/bench_decim_paths.py $c 2>&1 | tail -1 | awk -v i=$i -v c=$c '{print i, c, $5}' >> /tmp/var/samples.txt; done
    done
done
i=0
for flags in "$@"; do
    for c in $CFG; do
        echo "[$flags] $c: $(awk -v i=$i -v c=$c '$1==i && $2==c {print $3}' /tmp/var/samples.txt | tr '\n' ' ')" >> $OUT
    done
    i=$((i+1))
done
rm -f sdrdaemon_amd/csrc/build/decim_mfma.hip.o
make -s -C sdrdaemon_amd/csrc > /dev/null 2>&1
cat $OUT
