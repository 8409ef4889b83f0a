#!/bin/bash
# usage: ab_env.sh "<pytest -k expr or empty>" VAR OLD NEW  — tests, then same-box alternating 8-step bench runs with VAR=OLD / VAR=NEW
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
if [ -n "$1" ]; then timeout 1200 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -4; fi
VAR=$2; OLD=$3; NEW=$4
for rep in 1 2; do
  for v in old new; do
    VAL=$NEW; [ $v == old ] && VAL=$OLD
    env $VAR=$VAL timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass > $O/ab_${VAR}_${v}_$rep.json 2> $O/ab_${VAR}_${v}_$rep.err
    python - <<P
import json
try:
    d=json.loads(open("$O/ab_${VAR}_${v}_$rep.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$VAR=$VAL $rep", round(d["ms_per_step"],1), "region", round(r["region_ms_per_step"],2), "fwd", round(r["region_fwd_ms_per_step"],2), "bwd", round(r["region_bwd_ms_per_step"],2), "peak GB", round(d["config"].get("peak_hbm_gb",0),1))
except Exception as e:
    print("failed", e); print(open("$O/ab_${VAR}_${v}_$rep.err").read()[-1500:])
P
  done
done
