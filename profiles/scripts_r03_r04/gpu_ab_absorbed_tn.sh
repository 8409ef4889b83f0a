#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gemm_tn_gpu.py tests/test_sva_absorbed_gpu.py -m gpu -q -x > $O/pytest_tn.log 2>&1; tail -15 $O/pytest_tn.log )
X="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass --no-calibration --batch 16 --steps 2 --warmup 1"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" $X > "$GRAFT_REPO_ROOT/$O/prof.json" 2> "$GRAFT_REPO_ROOT/$O/prof.err" )
python tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) 80 > $O/kernel_stats.md 2>&1
rm -rf $O/prof
Y="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass --batch 16 --steps 8 --warmup 2"
( timeout 300 python3 bench.py $Y > $O/bench_tn.json 2> $O/bench_tn.err; echo "bench tn rc=$?" )
( CAMBRIAN_AMD_TN_WGRAD=0 timeout 300 python3 bench.py $Y > $O/bench_notn.json 2> $O/bench_notn.err; echo "bench notn rc=$?" )
( CAMBRIAN_AMD_ABSORB_KV=0 timeout 300 python3 bench.py $Y > $O/bench_noabs.json 2> $O/bench_noabs.err; echo "bench noabs rc=$?" )
python - <<'P'
import json
for f in ("bench_tn","bench_notn","bench_noabs"):
    try:
        d=json.load(open(f"gpurun_out/ab/{f}.json")); r=d.get("roofline",{})
        print(f, round(d["ms_per_step"],1), round(d["value"],3), "frac",round(r.get("frac",0),3), "region",round(r.get("region",{}).get("ms_per_step",0),1), round(r.get("region",{}).get("frac",0),3))
    except Exception as e: print(f, repr(e)[:300])
P
grep -E "sva_abs|gemm_nt_kernelIDF16b|gemm_tn|transpose_kernel|splitk" $O/kernel_stats.md | cut -c1-200
