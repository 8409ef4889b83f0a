#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's bench command -> per-kernel table (tools/rocpd_summary.py) + the line the
# profiled run printed
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/prof_driver; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$O/profiled_run.json" 2> "$GRAFT_REPO_ROOT/$O/profiled_run.err" )
python tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) 70 > $O/kernel_stats.md 2>&1
rm -rf $O/prof
head -12 $O/kernel_stats.md | cut -c1-180
python - <<'P'
import json
d=json.load(open("gpurun_out/prof_driver/profiled_run.json")); r=d["roofline"]
print(d["ms_per_step"], d["value"], r["frac"], r["avg_launch_us"] if "avg_launch_us" in r else {k:v for k,v in r.items() if "us" in k})
P
