#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag> [extra bench.py args, e.g. --math 2]   -> gpurun_out/<tag>/
# Every profiled process runs ONE arithmetic mode (bench.py measures the other modes only with --other-modes).
set -e
TAG=$1; shift
EXTRA="$@"
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT            # (gpurun merges a call's files into the local gpurun_out/: delete the local gpurun_out/<tag> before re-using a tag)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16 --detail $OUT/stats_detail.json $EXTRA"
# 1. plain bench (the reported line) + per-launch table
python3 $GRAFT_REPO_ROOT/bench.py --dump-launches $OUT/launches.csv --detail $OUT/bench_detail.json $EXTRA > $OUT/bench.json 2> $OUT/bench.err
# 2. kernel trace + stats of the same command.  Per-kernel durations and counters are taken on ONE stream: with bf16 tensors the
#    weight gradients otherwise run next to the dgrad chain (unet_set_overlap default) and every kernel's duration then includes its
#    neighbour's share of the machine (bench.py's own per-family events switch the overlap off for the steps they sample, too)
export UNET_OVERLAP=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
# 3. PMC passes (own runs, kernel-trace only)
PB="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-kernel-timing --detail $OUT/pmc_detail.json $EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -- $PB > $OUT/sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch --pmc FETCH_SIZE -- $PB > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/write --pmc WRITE_SIZE -- $PB > $OUT/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM -- $PB > $OUT/lds.log 2>&1
tail -c 600 $OUT/bench.json
