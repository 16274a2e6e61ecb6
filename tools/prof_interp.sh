#!/bin/bash
# GPU box: stall / busy counters of the VALU interpolator (one configuration of tools/bench_interp_paths.py)
#   -> gpurun_out/prof_interp2/summary.txt
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_interp2
rm -rf $OUT; mkdir -p $OUT
CMD="python $ROOT/tools/bench_interp_paths.py ${1:-valu:0:4}"
cd /tmp
run() {
    local d=$1; shift
    rocprofv3 --kernel-trace "$@" -d $OUT/$d -o run -- $CMD > $OUT/$d.log 2>&1
    python $ROOT/tools/rocpd_summary.py $(find $OUT/$d -name "*.db" | head -1) 2>&1 | sed "s#$OUT/##" > $OUT/$d.txt
    rm -rf $OUT/$d
}
run a --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run b --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR
# (TA_* / TCP_* counters crash rocprofv3 7.2 on this pool -- signal 6, then a hang until the time limit: left out)
{
    echo "# command: $CMD"
    for d in a b; do echo; echo "#### pass $d"; grep -A12 "PMC interp_" $OUT/$d.txt; grep "interp_" $OUT/$d.txt | head -1; done
} > $OUT/summary.txt
cat $OUT/summary.txt
