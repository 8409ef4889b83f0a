#!/bin/bash
# round-4 lab run 2: defaults after lab 1 (LayerNorm launch rule, prefetching column-walking dwconv, one GELU everywhere,
# one-barrier tower attention), 128-tile / TN GEMMs with and without the VGPR-form MFMA, kernel tests, same-box bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/r04_lab2.jsonl
timeout 600 python tools/r04_lab.py --out gpurun_out/r04_lab2.jsonl > gpurun_out/r04_lab2.log 2>&1; echo "lab rc=$?"
CAMBRIAN_AMD_LIB=$PWD/cambrian_amd/csrc/libcambrian_amd_novf.so timeout 300 python tools/r04_lab.py --only g128 --tag novf --out gpurun_out/r04_lab2.jsonl >> gpurun_out/r04_lab2.log 2>&1; echo "lab novf rc=$?"
timeout 900 python -m pytest tests/test_hook_link_gpu.py tests/test_kernels_gpu.py tests/test_gemm256_gpu.py tests/test_towers_gpu.py tests/test_sva_absorbed_gpu.py tests/test_sva_gpu.py -m gpu -x -q > gpurun_out/r04_lab2_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04_lab2_pytest.log
BF="--steps 6 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass"
CAMBRIAN_AMD_KNOBS="0=0,1=0,2=0" timeout 400 python bench.py $BF > gpurun_out/r04_ab2_knobs_off.json 2> gpurun_out/r04_ab2_knobs_off.err; echo "bench off rc=$?"
timeout 400 python bench.py $BF > gpurun_out/r04_ab2_knobs_on.json 2> gpurun_out/r04_ab2_knobs_on.err; echo "bench on rc=$?"
python - <<'PY'
import json
for f in ("off","on"):
    try:
        d=json.loads(open(f"gpurun_out/r04_ab2_knobs_{f}.json").read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("region",{}).get("frac"), r.get("region",{}))
    except Exception as e:
        print(f, "ERR", e)
PY
cat gpurun_out/r04_lab2.log | grep -v "^/opt/amdgpu" | tail -70
