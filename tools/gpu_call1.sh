#!/bin/bash
# round-3 GPU call 1: gpu tests, driver's bench command, call-site attribution, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c1/pytest.log )
tail -5 gpurun_out/c1/pytest.log
( timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c1/bench_driver.json 2> gpurun_out/c1/bench_driver.err; echo "bench rc=$?" )
tail -c 600 gpurun_out/c1/bench_driver.err
( timeout 300 python tools/profile_step_stacks.py > gpurun_out/c1/stacks.txt 2> gpurun_out/c1/stacks.err; echo "stacks rc=$?" )
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c1/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-masked-case --no-ab --no-gemm-pass > "$GRAFT_REPO_ROOT/gpurun_out/c1/bench_prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/c1/bench_prof.err"; echo "rocprof rc=$?" )
ls -la gpurun_out/c1/prof 2>/dev/null | head
python tools/rocpd_summary.py $(find gpurun_out/c1/prof -name "*.db" | head -1) > gpurun_out/c1/kernel_stats.md 2>&1
head -c 1500 gpurun_out/c1/bench_driver.json
