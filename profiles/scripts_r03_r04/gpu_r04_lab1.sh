#!/bin/bash
# round-4 lab run 1: kernel variants (tools/r04_lab.py), the two new GPU tests, same-box bench A/B of the knobs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/r04_lab.jsonl
timeout 600 python tools/r04_lab.py --out gpurun_out/r04_lab.jsonl > gpurun_out/r04_lab.log 2>&1; echo "lab rc=$?"
CAMBRIAN_AMD_LIB=$PWD/cambrian_amd/csrc/libcambrian_amd_novf.so timeout 300 python tools/r04_lab.py --only attn,flash --tag novf --out gpurun_out/r04_lab.jsonl >> gpurun_out/r04_lab.log 2>&1; echo "lab novf rc=$?"
timeout 600 python -m pytest tests/test_hook_link_gpu.py tests/test_kernels_gpu.py tests/test_gemm256_gpu.py -m gpu -x -q > gpurun_out/r04_lab1_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04_lab1_pytest.log
BF="--steps 6 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass"
timeout 400 python bench.py $BF > gpurun_out/r04_ab_knobs_off.json 2> gpurun_out/r04_ab_knobs_off.err; echo "bench off rc=$?"
CAMBRIAN_AMD_KNOBS="0=1,1=32,2=1,3=1" timeout 400 python bench.py $BF > gpurun_out/r04_ab_knobs_on.json 2> gpurun_out/r04_ab_knobs_on.err; echo "bench on rc=$?"
python - <<'PY'
import json
for f in ("off","on"):
    try:
        d=json.loads(open(f"gpurun_out/r04_ab_knobs_{f}.json").read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("region",{}).get("frac"), r.get("region",{}).get("ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat gpurun_out/r04_lab.log | tail -80
