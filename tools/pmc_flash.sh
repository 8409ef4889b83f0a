#!/bin/bash
# SQ counter passes over the decoder's flash attention kernels (tests/test_flash_bwd_gpu.py -k speed), each counter
# group in its own rocprofv3 run with --kernel-trace only.  Usage (GPU box, repo root): tools/pmc_flash.sh
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_flash
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python -m pytest $ROOT/tests/test_flash_bwd_gpu.py -q -k speed"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o sq2 -- $CMD > $OUT/sq2.log 2>&1
find $OUT -name "*.csv" | head
