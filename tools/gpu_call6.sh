#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c6; mkdir -p $O
export TMPDIR=/tmp
( timeout 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -8 $O/pytest.log
for i in 1 2 3; do
  ( timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_run$i.json 2> $O/bench_driver_run$i.err; echo "bench$i rc=$?" )
done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$O/bench_prof.json" 2> "$GRAFT_REPO_ROOT/$O/bench_prof.err"; echo "rocprof rc=$?" )
python tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.md 2>&1
rm -rf $O/prof
python - <<'P'
import json
for f in ("bench_driver_run1","bench_driver_run2","bench_driver_run3","bench_prof"):
    try:
        d=json.load(open(f"gpurun_out/c6/{f}.json")); r=d.get("roofline",{})
        print(f, round(d["ms_per_step"],1), round(d["value"],3), "B", d["config"]["images_per_gpu"], "frac",round(r.get("frac",0),3), "region",round(r.get("region",{}).get("ms_per_step",0),1), round(r.get("region",{}).get("frac",0),3), "gemm", round(r.get("all_own_gemm",{}).get("frac",0),3), "mem", round(d["config"].get("peak_hbm_gb"),1), round(d["config"].get("peak_hbm_reserved_gb",0),1), d.get("box",{}).get("rocm_smi",{}).get("sclk clock speed:"))
    except Exception as e: print(f, repr(e)[:300])
P
