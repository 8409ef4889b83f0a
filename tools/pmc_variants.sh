#!/bin/bash
# SQ counters of one GEMM shape for a list of tile hints (kernel variants), one rocprofv3 --pmc pass per variant.
# Usage (GPU box, repo root): tools/pmc_variants.sh M N K "2560 2600 2601" [act] -> gpurun_out/pmcv/summary.jsonl
set -u
M=$1; N=$2; K=$3; VARS=$4; ACT=${5:-0}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmcv
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $OUT/v$v -o v$v -- python $ROOT/tools/bench_one_gemm.py $M $N $K $v 3 $ACT > $OUT/v$v.log 2>&1
  python - $OUT/v$v $v <<'PY' >> $OUT/summary.jsonl
import csv, glob, json, sys, collections
d, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_nt" not in k and "Cijk" not in k:
            continue
        acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(json.dumps({"variant": v, "kernel": k, **{n: sum(x) / len(x) for n, x in c.items()}, "launches": len(next(iter(c.values())))}))
PY
  grep "TFLOP" $OUT/v$v.log >> $OUT/summary.jsonl
done
cat $OUT/summary.jsonl
