#!/bin/bash
# clocks and wait shares of K1m at one and two waves per SIMD (same spans)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/exp4
rm -rf $OUT; mkdir -p $OUT
run() { # <tag> <cmd...>
    local tag=$1; shift
    ( cd /tmp
    rocprofv3 --kernel-trace --stats -d $OUT/$tag.t -o run -- "$@" > $OUT/$tag.t.log 2>&1
    python $ROOT/tools/rocpd_summary.py $(find $OUT/$tag.t -name "*.db" | head -1) 2>&1 | grep -v "^==" | head -4 > $OUT/$tag.txt
    rm -rf $OUT/$tag.t
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/$tag.p -o run -- "$@" > $OUT/$tag.p.log 2>&1
    python $ROOT/tools/rocpd_summary.py $(find $OUT/$tag.p -name "*.db" | head -1) 2>&1 | grep -A12 "PMC decim_mfma\|decim_mfma_kernel" >> $OUT/$tag.txt
    rm -rf $OUT/$tag.p )
    echo "#### $tag: $@"; cat $OUT/$tag.txt
}
run s8 python $ROOT/tools/bench_decim_paths.py mfma:0:4 25 8
run s16 python $ROOT/tools/bench_decim_paths.py mfma:33792:4 25 16
run s8half python $ROOT/tools/bench_decim_paths.py mfma:16896:4 25 8
