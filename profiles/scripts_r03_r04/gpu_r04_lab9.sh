#!/bin/bash
# round-4 run 9: SwiGLU-pairs GEMM epilogue, token_mean / small LayerNorm backward launch shapes, calibration over single-round
# shapes; towers tests (DINOv2 uses the fused gate), step A/B is against run 7's numbers (different box: indicative only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py tests/test_towers_gpu.py tests/test_full_depth_gpu.py tests/test_release_dims_gpu.py -m gpu -x -q > gpurun_out/r04_lab9_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r04_lab9_pytest.log
BF="--steps 8 --warmup 3 --no-cpu-baseline --no-ab --no-masked-case"
timeout 600 python bench.py $BF --gemm-report gpurun_out/r04b_gemm_shapes_b24.json > gpurun_out/r04_ab9.json 2> gpurun_out/r04_ab9.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_ab9.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print(d["value"], d["ms_per_step"], r.get("frac"), {k:r["region"].get(k) for k in ("frac","ms_per_step","fwd_ms_per_step","bwd_ms_per_step","executed_tflop_per_step","executed_frac")}, r.get("all_own_gemm",{}).get("frac"))
for row in r.get("calibration",{}).get("shapes",[]): print(row["M"],row["N"],row["K"],row["act"],row["us"],row["choice"])
PY
