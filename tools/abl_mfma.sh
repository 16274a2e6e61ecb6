#!/bin/bash
# GPU box: timing-only ablations of the matrix-core decimator (make EXTRA=-DMF_ABL=n; results are wrong by design)
# usage: tools/abl_mfma.sh "0 1 2 4 8 15" [extra -D flags]   -> gpurun_out/abl_mfma.txt
ROOT=$PWD
OUT=$ROOT/gpurun_out/abl_mfma.txt
mkdir -p $ROOT/gpurun_out; : > $OUT
for abl in $1; do
    rm -f sdrdaemon_amd/csrc/build/decim_mfma.hip.o
    make -s -C sdrdaemon_amd/csrc EXTRA="-DMF_ABL=$abl $2" > /dev/null 2>&1 || { echo "build failed $abl" >> $OUT; continue; }
    echo "MF_ABL=$abl $2: $(python tools/bench_decim_paths.py mfma:0:4 2>&1 | tail -1)" >> $OUT
done
rm -f sdrdaemon_amd/csrc/build/decim_mfma.hip.o
make -s -C sdrdaemon_amd/csrc > /dev/null 2>&1
cat $OUT
