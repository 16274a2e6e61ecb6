#!/bin/bash
# usage (GPU box): tools/prof_cmd.sh <tag> <command...>   -> gpurun_out/prof_<tag>/{trace,pmc1,pmc2,pmc3,pmc4}
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="$@"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc1 -o run -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $OUT/pmc2 -o run -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o run -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o run -- $CMD > $OUT/pmc4.log 2>&1
