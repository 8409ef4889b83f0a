#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q > gpurun_out/c2/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c2/pytest.log )
tail -15 gpurun_out/c2/pytest.log
( timeout 400 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/c2/bench.json 2> gpurun_out/c2/bench.err; echo "bench rc=$?" )
( CAMBRIAN_AMD_NO_HOOK_LINK=1 timeout 400 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-ab --no-gemm-pass --no-masked-case > gpurun_out/c2/bench_nolink.json 2> gpurun_out/c2/bench_nolink.err; echo "bench nolink rc=$?" )
( timeout 300 python tools/profile_step_stacks.py > gpurun_out/c2/stacks.txt 2> gpurun_out/c2/stacks.err; echo "stacks rc=$?" )
python - <<'P'
import json
for f in ("bench","bench_nolink"):
    try:
        d=json.load(open(f"gpurun_out/c2/{f}.json"))
        r=d.get("roofline",{})
        print(f, round(d["ms_per_step"],1), round(d["value"],3), "frac",round(r.get("frac",0),3), "region",{k:round(v,2) for k,v in r.get("region",{}).items() if isinstance(v,float)})
    except Exception as e: print(f, e)
P
