#!/bin/bash
# round-4 lab run 3: p5 start-up stagger sweep, absorbed-kernel tests (exact instantiation), per-shape GEMM table of the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/r04_lab3.jsonl
timeout 600 python tools/r04_lab.py --only stagger --out gpurun_out/r04_lab3.jsonl > gpurun_out/r04_lab3.log 2>&1; echo "lab rc=$?"
timeout 900 python -m pytest tests/test_sva_absorbed_gpu.py tests/test_sva_gpu.py tests/test_kernels_gpu.py -m gpu -x -q > gpurun_out/r04_lab3_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r04_lab3_pytest.log
timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --gemm-report gpurun_out/r04_gemm_shapes_b24.json > gpurun_out/r04_gemm_report_run.json 2> gpurun_out/r04_gemm_report_run.err; echo "bench rc=$?"
cat gpurun_out/r04_lab3.log | grep -v "^/opt/amdgpu" | tail -60
