#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (each in its own rocprofv3 run, --kernel-trace only) for ONE case of
# tools/bench_hbm_kernels.py.  Usage (GPU box, repo root): tools/pmc_hbm.sh "sva_fwd" tag  -> gpurun_out/pmc_hbm/<tag>/{fetch,write}
# Per launch: HBM bytes = 2 x FETCH_SIZE x 1024 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md §HBM) + WRITE_SIZE x 1024.
set -u
CASE="$1"; TAG="$2"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_hbm/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_hbm_kernels.py --iters 5 --only"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD "$CASE" > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD "$CASE" > $OUT/write.log 2>&1
find $OUT -name "*counter_collection.csv" | head
