#!/bin/bash
# GPU box: counters of the matrix-core decimator kernel (one configuration of tools/bench_decim_paths.py)
# usage: tools/prof_mfma.sh [path:span:log2decim]   -> gpurun_out/prof_mfma/summary.txt
export TMPDIR=/tmp
ROOT=$PWD
CFG=${1:-mfma:0:4}
OUT=$ROOT/gpurun_out/prof_mfma
rm -rf $OUT; mkdir -p $OUT
CMD="python $ROOT/tools/bench_decim_paths.py $CFG"
cd /tmp
run() {
    local d=$1; shift
    rocprofv3 --kernel-trace "$@" -d $OUT/$d -o run -- $CMD > $OUT/$d.log 2>&1
    python $ROOT/tools/rocpd_summary.py $(find $OUT/$d -name "*.db" | head -1) 2>&1 | sed "s#$OUT/##" > $OUT/$d.txt
    rm -rf $OUT/$d
}
run trace --stats
run pmc1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS
run pmc2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run pmc3 --pmc FETCH_SIZE
run pmc4 --pmc WRITE_SIZE
run pmc5 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES
run pmc6 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL
{
    echo "# command: $CMD"
    for d in trace pmc1 pmc2 pmc3 pmc4 pmc5 pmc6; do echo; echo "#### pass $d"; cat $OUT/$d.txt; done
} > $OUT/summary.txt
