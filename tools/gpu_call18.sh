#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c18; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_sva_absorbed_gpu.py -m gpu -q -x > $O/pytest_abs.log 2>&1; tail -4 $O/pytest_abs.log )
X="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass --no-calibration --batch 16 --steps 2 --warmup 1"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_abs" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" $X > "$GRAFT_REPO_ROOT/$O/abs.json" 2> "$GRAFT_REPO_ROOT/$O/abs.err" )
python tools/rocpd_summary.py $(find $O/prof_abs -name "*.db" | head -1) 80 > $O/kernel_stats_abs.md 2>&1
rm -rf $O/prof_abs
Y="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass --batch 16 --steps 8 --warmup 2"
( timeout 300 python3 bench.py $Y > $O/bench_abs.json 2> $O/bench_abs.err; echo "bench abs rc=$?" )
( CAMBRIAN_AMD_ABSORB_KV=0 timeout 300 python3 bench.py $Y > $O/bench_noabs.json 2> $O/bench_noabs.err; echo "bench noabs rc=$?" )
python - <<'P'
import json
for f in ("bench_abs","bench_noabs"):
    try:
        d=json.load(open(f"gpurun_out/c18/{f}.json")); r=d.get("roofline",{})
        print(f, round(d["ms_per_step"],1), round(d["value"],3), "frac",round(r.get("frac",0),3), "region",round(r.get("region",{}).get("ms_per_step",0),1), round(r.get("region",{}).get("frac",0),3))
    except Exception as e: print(f, repr(e)[:300])
P
grep -E "sva_abs|gemm_nt_kernelIDF16b|transpose_kernel|reduce_kernel" $O/kernel_stats_abs.md | cut -c1-200
