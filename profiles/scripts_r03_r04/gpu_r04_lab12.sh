#!/bin/bash
# round-4 lab run 12: the leaner GELU epilogue (degree-5 polynomial, the four pairs' chains interleaved, accumulator reads in
# one block) against the previous library (a lab build or the parent commit: cambrian_amd/csrc/libcambrian_amd_lab_prev.so), per shape and on
# the whole step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/r04_lab15.jsonl
OLD=$PWD/cambrian_amd/csrc/libcambrian_amd_lab_prev.so
timeout 900 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py tests/test_towers_gpu.py -m gpu -x -q > gpurun_out/r04_lab15_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04_lab15_pytest.log
for i in 1 2; do
timeout 300 python tools/r04_lab.py --only gelu,epi --out gpurun_out/r04_lab15.jsonl --tag new$i > gpurun_out/r04_lab15.log 2>&1; echo "lab new rc=$?"
CAMBRIAN_AMD_LIB=$OLD timeout 300 python tools/r04_lab.py --only gelu,epi --out gpurun_out/r04_lab15.jsonl --tag old$i >> gpurun_out/r04_lab15.log 2>&1; echo "lab old rc=$?"
done
BF="--steps 6 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass"
for i in 1 2; do
CAMBRIAN_AMD_LIB=$OLD timeout 400 python bench.py $BF > gpurun_out/r04_ab15_old$i.json 2> gpurun_out/r04_ab15_old$i.err; echo "bench old rc=$?"
timeout 400 python bench.py $BF > gpurun_out/r04_ab15_new$i.json 2> gpurun_out/r04_ab15_new$i.err; echo "bench new rc=$?"
done
python - <<'PY'
import json
for f in ("old1","new1","old2","new2"):
    try:
        d=json.loads(open(f"gpurun_out/r04_ab15_{f}.json").read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("region",{}).get("frac"), r.get("region",{}).get("ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
rows=[json.loads(l) for l in open("gpurun_out/r04_lab15.jsonl")]
by={}
for r in rows:
    if "us" in r: by.setdefault((r["kernel"],r["shape"]),{})[r["tag"]]=r["us"]
for k,v in by.items():
    print(f"{str(k):80s}", "  ".join(f"{t} {v[t]:8.1f}" for t in sorted(v)))
for r in rows:
    if "check" in r and r["tag"]=="new1": print(r)
PY
