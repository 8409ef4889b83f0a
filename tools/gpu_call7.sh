#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c7; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_sva_absorbed_gpu.py -m gpu -q -x > $O/pytest_abs.log 2>&1; echo "pytest abs rc=$?" >> $O/pytest_abs.log )
tail -30 $O/pytest_abs.log
( timeout 900 python -m pytest tests/test_sva_gpu.py tests/test_model_gpu.py tests/test_release_dims_gpu.py tests/test_dynamic_gpu.py tests/test_gemm256_gpu.py tests/test_kernels_gpu.py -m gpu -q > $O/pytest_more.log 2>&1; echo "pytest more rc=$?" >> $O/pytest_more.log )
tail -12 $O/pytest_more.log
X="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass --batch 16"
( timeout 300 python3 bench.py --steps 8 --warmup 2 $X > $O/bench_abs.json 2> $O/bench_abs.err; echo "bench abs rc=$?" )
( CAMBRIAN_AMD_ABSORB_KV=0 timeout 300 python3 bench.py --steps 8 --warmup 2 $X > $O/bench_noabs.json 2> $O/bench_noabs.err; echo "bench noabs rc=$?" )
python - <<'P'
import json
for f in ("bench_abs","bench_noabs"):
    try:
        d=json.load(open(f"gpurun_out/c7/{f}.json")); r=d.get("roofline",{})
        print(f, round(d["ms_per_step"],1), round(d["value"],3), "loss", d["config"]["loss"], "frac",round(r.get("frac",0),3), "region",round(r.get("region",{}).get("ms_per_step",0),1), round(r.get("region",{}).get("frac",0),3), "mem", round(d["config"].get("peak_hbm_gb"),1))
    except Exception as e: print(f, repr(e)[:300])
P
tail -5 $O/bench_abs.err
