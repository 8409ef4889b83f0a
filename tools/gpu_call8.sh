#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c8; mkdir -p $O
export TMPDIR=/tmp
X="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass --no-calibration --batch 16 --steps 2 --warmup 1"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_abs" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" $X > "$GRAFT_REPO_ROOT/$O/abs.json" 2> "$GRAFT_REPO_ROOT/$O/abs.err" )
python tools/rocpd_summary.py $(find $O/prof_abs -name "*.db" | head -1) 80 > $O/kernel_stats_abs.md 2>&1
rm -rf $O/prof_abs
( cd /tmp && CAMBRIAN_AMD_ABSORB_KV=0 timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_no" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" $X > "$GRAFT_REPO_ROOT/$O/no.json" 2> "$GRAFT_REPO_ROOT/$O/no.err" )
python tools/rocpd_summary.py $(find $O/prof_no -name "*.db" | head -1) 80 > $O/kernel_stats_no.md 2>&1
rm -rf $O/prof_no
( timeout 300 python -m pytest tests/test_sva_absorbed_gpu.py -m gpu -q > $O/pytest_abs.log 2>&1; tail -3 $O/pytest_abs.log )
