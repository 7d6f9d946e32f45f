#!/bin/bash
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> ; separate passes, kernel-trace only (no sys-trace)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -- $B > $OUT/sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch --pmc FETCH_SIZE -- $B > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/write --pmc WRITE_SIZE -- $B > $OUT/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/lds --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA -- $B > $OUT/lds.log 2>&1
find $OUT -name "*.csv" | head -20
