#!/bin/bash
# L2 / fabric counters of one GEMM shape for a list of tile hints: FETCH_SIZE, WRITE_SIZE, TCC hit / miss, each group in
# its own rocprofv3 --pmc pass (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots).
# Usage (GPU box, repo root): tools/pmc_l2.sh M N K "2560 2590 blas" -> gpurun_out/pmcl2/summary.jsonl
set -u
M=$1; N=$2; K=$3; VARS=$4; ACT=${5:-0}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmcl2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    tag=$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/v$v/$tag -o p -- python $ROOT/tools/bench_one_gemm.py $M $N $K $v 3 $ACT > $OUT/v$v.$tag.log 2>&1
  done
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
    print(json.dumps({"variant": v, "kernel": k, **{n: sum(x) / len(x) for n, x in c.items()}}))
PY
done
cat $OUT/summary.jsonl
grep -il "error\|invalid\|not found" $OUT/*.log | head
