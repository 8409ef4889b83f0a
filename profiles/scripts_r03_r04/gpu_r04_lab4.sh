#!/bin/bash
# round-4 lab run 4: the p5 epilogue with compile-time bias / LayerScale / residual flags against the previous epilogue (a
# second library with only gemm_p5.o rebuilt from the parent commit: CAMBRIAN_AMD_LIB), per shape and on the whole step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/r04_lab4.jsonl
OLD=$PWD/cambrian_amd/csrc/libcambrian_amd_oldepi.so
timeout 600 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > gpurun_out/r04_lab4_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04_lab4_pytest.log
timeout 300 python tools/r04_lab.py --only gelu,epi --out gpurun_out/r04_lab4.jsonl --tag new > gpurun_out/r04_lab4.log 2>&1; echo "lab new rc=$?"
CAMBRIAN_AMD_LIB=$OLD timeout 300 python tools/r04_lab.py --only gelu,epi --out gpurun_out/r04_lab4.jsonl --tag old >> gpurun_out/r04_lab4.log 2>&1; echo "lab old rc=$?"
BF="--steps 6 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass"
CAMBRIAN_AMD_LIB=$OLD timeout 400 python bench.py $BF > gpurun_out/r04_ab4_old.json 2> gpurun_out/r04_ab4_old.err; echo "bench old rc=$?"
timeout 400 python bench.py $BF > gpurun_out/r04_ab4_new.json 2> gpurun_out/r04_ab4_new.err; echo "bench new rc=$?"
CAMBRIAN_AMD_LIB=$OLD timeout 400 python bench.py $BF > gpurun_out/r04_ab4_old2.json 2> gpurun_out/r04_ab4_old2.err; echo "bench old rc=$?"
timeout 400 python bench.py $BF > gpurun_out/r04_ab4_new2.json 2> gpurun_out/r04_ab4_new2.err; echo "bench new rc=$?"
python - <<'PY'
import json
for f in ("old","new","old2","new2"):
    try:
        d=json.loads(open(f"gpurun_out/r04_ab4_{f}.json").read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("region",{}).get("frac"), r.get("region",{}).get("ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r04_lab4.jsonl")]
by={}
for r in rows:
    if "us" in r: by.setdefault(r["shape"],{})[r["tag"]]=r
for k,v in by.items():
    if "old" in v and "new" in v: print(f"{k:60s} old {v['old']['us']:8.1f} us {v['old']['tflops']:7.1f}  new {v['new']['us']:8.1f} us {v['new']['tflops']:7.1f}  x{v['old']['us']/v['new']['us']:.3f}")
for r in rows:
    if "check" in r: print(r)
PY
