#!/bin/bash
# GPU box: build variants of the VALU interpolator (EXTRA flags per argument, "-" = none), interleaved timing rounds
#   -> gpurun_out/var_interp.txt
ROOT=$PWD
OUT=$ROOT/gpurun_out/var_interp.txt
ROUNDS=${ROUNDS:-3}
export REPS=${REPS:-40}
mkdir -p $ROOT/gpurun_out /tmp/var; : > $OUT; rm -f /tmp/var/ik_samples.txt
i=0
for flags in "$@"; do
    [ "$flags" = "-" ] && flags=""
    rm -f sdrdaemon_amd/csrc/build/interp_kernels.hip.o sdrdaemon_amd/csrc/build/interp_mfma.hip.o
    if make -s -C sdrdaemon_amd/csrc EXTRA="$flags" > /tmp/var/build_$i.log 2>&1; then cp sdrdaemon_amd/libsdrhip.so /tmp/var/lib_$i.so; else echo "build failed: $flags" >> $OUT; fi
    i=$((i+1))
done
n=$i
for r in $(seq $ROUNDS); do
    for i in $(seq 0 $((n-1))); do
        [ -f /tmp/var/lib_$i.so ] || continue
        cp /tmp/var/lib_$i.so sdrdaemon_amd/libsdrhip.so
        python tools/bench_interp_paths.py valu:0:4 2>&1 | tail -1 | awk -v i=$i '{print i, $5, $NF}' >> /tmp/var/ik_samples.txt
    done
done
i=0
for flags in "$@"; do
    echo "[$flags] interpolate16 ms: $(awk -v i=$i '$1==i {print $2}' /tmp/var/ik_samples.txt | sort -n | tr '\n' ' ')" >> $OUT
    i=$((i+1))
done
rm -f sdrdaemon_amd/csrc/build/interp_kernels.hip.o sdrdaemon_amd/csrc/build/interp_mfma.hip.o
make -s -C sdrdaemon_amd/csrc > /dev/null 2>&1
cat $OUT
