#!/bin/bash
# round-4 lab run 14: bias / LayerScale lane-half selects as bit merges, against the previous library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OLD=$PWD/cambrian_amd/csrc/libcambrian_amd_lab_prev.so
timeout 600 python -m pytest tests/test_gemm256_gpu.py tests/test_towers_gpu.py -m gpu -x -q > gpurun_out/r04_lab20_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04_lab20_pytest.log
BF="--steps 6 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass"
for i in 1 2; do
CAMBRIAN_AMD_LIB=$OLD timeout 400 python bench.py $BF > gpurun_out/r04_ab20_old$i.json 2> gpurun_out/r04_ab20_old$i.err; echo "bench old rc=$?"
timeout 400 python bench.py $BF > gpurun_out/r04_ab20_new$i.json 2> gpurun_out/r04_ab20_new$i.err; echo "bench new rc=$?"
done
python - <<'PY'
import json
for f in ("old1","new1","old2","new2"):
    d=json.loads(open(f"gpurun_out/r04_ab20_{f}.json").read().strip().splitlines()[-1]); r=d.get("roofline",{})
    print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("region",{}).get("frac"), r.get("region",{}).get("ms_per_step"))
PY
