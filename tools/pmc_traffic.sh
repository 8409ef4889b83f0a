#!/bin/bash
# PMC passes for the dominant kernel (the 256x256 bf16 GEMM), each counter group in its OWN rocprofv3 run with
# --kernel-trace only (MI355X_MICROARCH.md §rocprofv3 PMC slots: FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2).
# Usage (on the GPU box, from the repo root): tools/pmc_traffic.sh M N K [tile_hint] [act]  -> gpurun_out/pmc/{sq,sq2,fetch,write}/...
# then (anywhere): python tools/pmc_summarise.py gpurun_out/pmc M N K profiles/rNN_pmc_<kernel>.json
set -u
M=${1:-32768}; N=${2:-6144}; K=${3:-1536}; TILE=${4:-0}; ACT=${5:-0}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_one_gemm.py $M $N $K $TILE 3 $ACT"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
tail -2 $OUT/sq.log $OUT/fetch.log
