#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py tests/test_towers_gpu.py -m gpu -x -q > gpurun_out/r04_lab10_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r04_lab10_pytest.log
BF="--steps 8 --warmup 3 --no-cpu-baseline --no-ab --no-masked-case"
timeout 600 python bench.py $BF --gemm-report gpurun_out/r04c_gemm_shapes_b24.json > gpurun_out/r04_ab10.json 2> gpurun_out/r04_ab10.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_ab10.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print(d["value"], d["ms_per_step"], r.get("frac"), {k:r["region"].get(k) for k in ("frac","ms_per_step","fwd_ms_per_step","bwd_ms_per_step","executed_tflop_per_step","executed_frac")}, r.get("all_own_gemm",{}).get("frac"), r.get("all_own_gemm",{}).get("ms_per_step"))
rows=json.load(open("gpurun_out/r04c_gemm_shapes_b24.json"))
for x in rows[:28]: print(f"{x['M']:>8} {x['N']:>6} {x['K']:>6} {x['act']:>2} k{x['kernel']:>5} n{x['launches_per_step']:>5.0f} {x['ms_per_step']:>6.2f} ms {x['TFLOPs']:>6.0f}")
PY
