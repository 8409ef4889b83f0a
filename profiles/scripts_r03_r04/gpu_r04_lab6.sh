#!/bin/bash
# round-4 lab run 6: deferred multi-layer LayerNorm backward (cmb_layernorm_bwd_multi): kernel + model tests, same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sva_gpu.py tests/test_sva_absorbed_gpu.py tests/test_model_gpu.py tests/test_release_dims_gpu.py -m gpu -x -q > gpurun_out/r04_lab6_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r04_lab6_pytest.log
BF="--steps 6 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass"
for i in 1 2; do
CAMBRIAN_AMD_DEFER_LN_BWD=0 timeout 400 python bench.py $BF > gpurun_out/r04_ab6_off$i.json 2> gpurun_out/r04_ab6_off$i.err; echo "bench off rc=$?"
timeout 400 python bench.py $BF > gpurun_out/r04_ab6_on$i.json 2> gpurun_out/r04_ab6_on$i.err; echo "bench on rc=$?"
done
python - <<'PY'
import json
for f in ("off1","on1","off2","on2"):
    try:
        d=json.loads(open(f"gpurun_out/r04_ab6_{f}.json").read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("region",{}).get("frac"), r.get("region",{}).get("ms_per_step"), r.get("region",{}).get("bwd_ms_per_step"), d.get("peak_mem_gb", d.get("config",{})))
    except Exception as e:
        print(f, "ERR", e)
PY
