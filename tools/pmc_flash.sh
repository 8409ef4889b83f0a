#!/bin/bash
# SQ counter passes for the decoder attention kernels (tools/r04_lab.py --only flash: 8 x 2048 tokens, 32 / 8 heads), --kernel-trace only.
# Usage (GPU box, repo root): tools/pmc_flash.sh -> gpurun_out/pmc_flash/{sq,sq2}; python tools/pmc_sq_summarise.py gpurun_out/pmc_flash <kernel>
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_flash
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/r04_lab.py --only flash --iters 3 --out /tmp/pmc_flash.jsonl"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o sq2 -- $CMD > $OUT/sq2.log 2>&1
