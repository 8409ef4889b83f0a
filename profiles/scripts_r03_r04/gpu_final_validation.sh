#!/bin/bash
# round-end check on one MI355X box: the whole -m gpu suite, smoke(), and the driver's bench command twice
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log )
( timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log )
for i in 1 2; do
  ( timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_run$i.json 2> $O/bench_driver_run$i.err; echo "bench $i rc=$?" )
done
python - <<'P'
import json
for i in (1, 2):
    try:
        d=json.load(open(f"gpurun_out/final/bench_driver_run{i}.json")); r=d.get("roofline",{})
        print("driver", i, d["config"].get("global_batch"), round(d["ms_per_step"],1), round(d["value"],3), "frac",round(r.get("frac",0),3), "region",round(r.get("region",{}).get("ms_per_step",0),1), round(r.get("region",{}).get("frac",0),3), "all_own", round(r.get("all_own_gemm",{}).get("frac",0),3))
    except Exception as e: print(repr(e)[:300])
P
