#!/bin/bash
# SQ counter pass (issue / wait / VALU activity) for ONE case of tools/bench_hbm_kernels.py, --kernel-trace only.
# Usage (GPU box, repo root): tools/pmc_sq.sh "dwconv7x7 [stage 3" tag  -> gpurun_out/pmc_sq/<tag>/sq/...; summarise with
#   python tools/pmc_sq_summarise.py gpurun_out/pmc_sq/<tag> <kernel substring>
set -u
CASE="$1"; TAG="$2"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_sq/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_hbm_kernels.py --iters 5 --only"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq -o sq -- $CMD "$CASE" > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o sq2 -- $CMD "$CASE" > $OUT/sq2.log 2>&1
find $OUT -name "*counter_collection.csv" | head
