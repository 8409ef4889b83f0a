#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python tools/r04_lab.py --only lnmulti --out gpurun_out/r04_lab7.jsonl > gpurun_out/r04_lab7.log 2>&1; echo "lab rc=$?"; grep -v amdgpu gpurun_out/r04_lab7.log | tail
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_sva_gpu.py -m gpu -x -q -k "layernorm or sva" > gpurun_out/r04_lab7_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04_lab7_pytest.log
BF="--steps 6 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass"
CAMBRIAN_AMD_DEFER_LN_BWD=0 timeout 400 python bench.py $BF > gpurun_out/r04_ab7_off.json 2> gpurun_out/r04_ab7_off.err; echo "bench off rc=$?"
timeout 400 python bench.py $BF > gpurun_out/r04_ab7_on7.json 2> gpurun_out/r04_ab7_on7.err; echo "bench on rc=$?"
CAMBRIAN_AMD_KNOBS="4=4" timeout 400 python bench.py $BF > gpurun_out/r04_ab7_on4.json 2> gpurun_out/r04_ab7_on4.err; echo "bench on4 rc=$?"
python - <<'PY'
import json
for f in ("off","on7","on4"):
    try:
        d=json.loads(open(f"gpurun_out/r04_ab7_{f}.json").read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("region",{}).get("frac"), r.get("region",{}).get("ms_per_step"), r.get("region",{}).get("bwd_ms_per_step"), d["config"].get("peak_hbm_gb"))
    except Exception as e:
        print(f, "ERR", e)
PY
