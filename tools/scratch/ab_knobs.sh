#!/bin/bash
# same-box A/B of the round-6 knob defaults: alternating 8-step bench runs, region ms per step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do
  for v in old new; do
    K=""; [ $v == old ] && K="6=0,7=16"
    CAMBRIAN_AMD_KNOBS=$K timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass > $O/ab_knobs_${v}_$rep.json 2> $O/ab_knobs_${v}_$rep.err
    python - <<P
import json
d=json.loads(open("$O/ab_knobs_${v}_$rep.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$v $rep", round(d["ms_per_step"],1), "region", round(r["region_ms_per_step"],2), "fwd", round(r["region_fwd_ms_per_step"],2), "bwd", round(r["region_bwd_ms_per_step"],2))
P
  done
done
