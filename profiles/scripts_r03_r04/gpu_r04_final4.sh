#!/bin/bash
# round-4 final validation: PMC passes of the dominant kernel on its hottest launch shape (first, so that the bench lines
# carry roofline.traffic), the driver's command twice, its rocprofv3 kernel table, the per-shape GEMM table, the whole GPU
# test suite and smoke()
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -rf gpurun_out/pmc; bash tools/pmc_traffic.sh 98304 6144 1536 2590 1 > gpurun_out/r04d_pmc_traffic.log 2>&1; echo "pmc rc=$?"
python tools/pmc_summarise.py gpurun_out/pmc 98304 6144 1536 gpurun_out/r04_pmc_gemm_p5.json cambrian_amd/csrc/gemm_p5.hip cambrian_amd/csrc/gemm_p5_epilogue.inc && cp gpurun_out/r04_pmc_gemm_p5.json profiles/r04_pmc_gemm_p5.json
rm -rf gpurun_out/pmc/*/*.db 2>/dev/null
for i in 1 2; do
  timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04d_bench_driver_cmd_run$i.json 2> gpurun_out/r04d_bench_driver_cmd_run$i.err; echo "driver cmd run $i rc=$?"
done
bash tools/gpu_profile_driver_cmd.sh > gpurun_out/r04d_profile.log 2>&1; echo "profile rc=$?"
cp gpurun_out/prof_driver/kernel_stats.md gpurun_out/r04d_bench_b24_kernel_stats.md; cp gpurun_out/prof_driver/profiled_run.json gpurun_out/r04d_bench_b24_profiled_run.json
timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --gemm-report gpurun_out/r04d_gemm_shapes_b24.json > gpurun_out/r04d_gemm_report_run.json 2> gpurun_out/r04d_gemm_report_run.err; echo "gemm report rc=$?"
timeout 1500 python -m pytest tests/ -m gpu -x -q > gpurun_out/r04d_pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04d_pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04d_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04d_smoke.log
python - <<'PY'
import json
for f in ("r04d_bench_driver_cmd_run1","r04d_bench_driver_cmd_run2","r04d_bench_b24_profiled_run","r04d_gemm_report_run"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("traffic"), {k:r.get("region",{}).get(k) for k in ("frac","ms_per_step","executed_frac")}, (r.get("all_own_gemm") or {}).get("frac"), d["config"].get("peak_hbm_gb"))
    except Exception as e:
        print(f, "ERR", e)
PY
