#!/bin/bash
# usage (GPU box, from the repo root): tools/prof_cmd.sh <tag> <command with absolute paths...>
# kernel trace + separate PMC passes of the command; condensed summaries -> gpurun_out/prof_<tag>/<pass>.txt
# (the rocpd databases are dropped: too big to travel back) and all of them -> gpurun_out/prof_<tag>/summary.txt
export TMPDIR=/tmp
ROOT=$PWD
TAG=$1; shift
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="$@"
cd /tmp
run() { # <pass> <rocprofv3 options...>
    local d=$1; shift
    rocprofv3 --kernel-trace "$@" -d $OUT/$d -o run -- $CMD > $OUT/$d.log 2>&1
    python $ROOT/tools/rocpd_summary.py $(find $OUT/$d -name "*.db" | head -1) 2>&1 | sed "s#$OUT/##" > $OUT/$d.txt
    rm -rf $OUT/$d
}
run trace --stats
run pmc1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run pmc2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run pmc3 --pmc FETCH_SIZE
run pmc4 --pmc WRITE_SIZE
{
    echo "# command: $CMD"
    echo "# tools/prof_cmd.sh on MI355X: rocprofv3 --kernel-trace --stats, then separate --pmc passes; condensed by tools/rocpd_summary.py"
    echo "# FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE needs the x2 correction on gfx950)"
    for d in trace pmc1 pmc2 pmc3 pmc4; do echo; echo "#### pass $d"; cat $OUT/$d.txt; done
} > $OUT/summary.txt
