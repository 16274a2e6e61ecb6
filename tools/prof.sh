#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the default bench command;
# condensed summaries -> gpurun_out/prof/<pass>.txt (the rocpd databases are dropped: too big to travel back)
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
CMD="python $ROOT/bench.py --cpu-seconds 0 $BENCH_ARGS"
cd /tmp
run() { # <pass> <rocprofv3 options...>
    local d=$1; shift
    rocprofv3 --kernel-trace "$@" -d $OUT/$d -o run -- $CMD > $OUT/$d.log 2>&1
    python $ROOT/tools/rocpd_summary.py $(find $OUT/$d -name "*.db" | head -1) > $OUT/$d.txt 2>&1
    rm -rf $OUT/$d
}
run trace --stats
run pmc1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run pmc2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run pmc3 --pmc FETCH_SIZE
run pmc4 --pmc WRITE_SIZE
tail -c 400 $OUT/trace.log > $OUT/trace_tail.log; rm -f $OUT/*.log.full
ls -la $OUT
