#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/unfrozen; mkdir -p $O
export TMPDIR=/tmp
X="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass --no-calibration --unfreeze-towers --batch 4 --steps 3 --warmup 1"
( timeout 400 python3 bench.py $X > $O/bench_unfrozen.json 2> $O/bench_unfrozen.err; echo "rc=$?"; tail -3 $O/bench_unfrozen.err )
python - <<'P'
import json
try:
    d=json.load(open("gpurun_out/unfrozen/bench_unfrozen.json")); r=d.get("roofline",{})
    print(d["config"].get("global_batch"), round(d["ms_per_step"],1), round(d["value"],3), d["config"].get("peak_hbm_gb"), d["config"].get("trainable_parameters"))
except Exception as e: print(repr(e)[:300])
P
