#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c5; mkdir -p $O
export TMPDIR=/tmp
X="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass"
for b in 16 20 24; do
  ( timeout 300 python3 bench.py --batch $b --steps 6 --warmup 2 $X > $O/bench_b$b.json 2> $O/bench_b$b.err; echo "b$b rc=$?" )
done
( timeout 120 python tools/bench_hbm_kernels.py --only layernorm_fwd > $O/ln.md 2>&1 )
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_towers_gpu.py tests/test_release_width_gpu.py -m gpu -q -k "layernorm or trunk or release_width and bf16" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -4 $O/pytest.log
cat $O/ln.md
python - <<'P'
import json
for b in (16,20,24):
    try:
        d=json.load(open(f"gpurun_out/c5/bench_b{b}.json")); r=d.get("roofline",{})
        print(b, round(d["ms_per_step"],1), round(d["value"],3), "frac",round(r.get("frac",0),3), "region",round(r.get("region",{}).get("ms_per_step",0),1), round(r.get("region",{}).get("frac",0),3), "mem", round(d["config"].get("peak_hbm_gb"),1))
    except Exception as e: print(b, repr(e)[:300])
P
tail -3 $O/bench_b24.err
