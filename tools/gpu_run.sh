#!/bin/bash
# ONE parametrised gpurun command file (rounds 5+; no per-call scratch scripts).  ROUND=r06 (default) names the output folder.
#   tools/gpu_run.sh bench            the driver's command, line -> gpurun_out/r05/bench_<tag>.json
#   tools/gpu_run.sh prof [steps]     rocprofv3 --kernel-trace --stats of the driver's command -> kernel_stats.md + timeline.json
#   tools/gpu_run.sh tests [expr]     pytest -m gpu (optionally -k expr) + smoke
#   tools/gpu_run.sh shapes           bench.py --gemm-report: per-shape GEMM table of a step
#   tools/gpu_run.sh lab <args...>    python tools/r05_lab.py <args...>
#   tools/gpu_run.sh py <file> ...    python <file> ...
# several verbs in one call: separate with '--'  (e.g. `tools/gpu_run.sh bench -- prof`)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${ROUND:-r06}; mkdir -p $O
TAG=${TAG:-$(date +%H%M%S)}
run_verb() {
  local verb=$1; shift
  case "$verb" in
    bench)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $O/bench_$TAG.json 2> $O/bench_$TAG.err
      echo "bench rc=$? $(wc -c < $O/bench_$TAG.json) bytes"; cp gpurun_out/bench_line_full.json $O/bench_full_$TAG.json 2>/dev/null
      python - <<P
import json
d=json.loads(open("$O/bench_$TAG.json").read().strip().splitlines()[-1]); r=d.get("roofline",{})
print("ms/step", round(d["ms_per_step"],1), "img/s", round(d["value"],3), "frac", r.get("frac"), "region", r.get("region_ms_per_step"), r.get("region_frac"), "own_gemm", r.get("all_own_gemm_frac"))
P
      ;;
    prof)
      local steps=${1:-20}
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps $steps --warmup 5 --no-cpu-baseline --no-ab > "$GRAFT_REPO_ROOT/$O/profiled_run_$TAG.json" 2> "$GRAFT_REPO_ROOT/$O/profiled_run_$TAG.err" )
      DB=$(find $O/prof -name "*.db" | head -1)
      python tools/rocpd_summary.py $DB 70 > $O/kernel_stats_$TAG.md 2>&1
      python tools/rocpd_timeline.py $DB > $O/timeline_$TAG.json 2> $O/timeline_$TAG.err
      rm -rf $O/prof
      head -14 $O/kernel_stats_$TAG.md | cut -c1-170; head -c 1500 $O/timeline_$TAG.json
      ;;
    tests)
      if [ -n "$1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -15
      else timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15; fi
      python __graft_entry__.py smoke 2>&1 | tail -2
      ;;
    modes)   # the NOT_HEADLINE lines: finetune stage with fp32 masters (plain / ZeRO-2 / ZeRO-3 units at world 1), fp8 projections
      for m in "--stage finetune" "--stage finetune --zero2" "--stage finetune --zero3" "--fp8-projections"; do
        n=$(echo "$m" | tr -d ' -'); 
        timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass $m > $O/mode_$n.json 2> $O/mode_$n.err
        echo "$m rc=$?"; python - <<P
import json
try:
    d=json.loads(open("$O/mode_$n.json").read().strip().splitlines()[-1]); c=d["config"]
    print(round(d["ms_per_step"],1), "ms", round(d["value"],3), "img/s", "peak", round(c.get("peak_hbm_gb",0),1), "GB", c.get("images_per_gpu"), c.get("optimizer_state"), c.get("NOT_HEADLINE","")[:60])
except Exception as e: print("parse failed", e); print(open("$O/mode_$n.err").read()[-1500:])
P
      done
      ;;
    shapes)  # per-shape table of every own GEMM launch of a step (an event pair per launch) -> gemm_shapes_<tag>.json
      timeout 500 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --gemm-report $O/gemm_shapes_$TAG.json > $O/gemm_report_run_$TAG.json 2> $O/gemm_report_run_$TAG.err
      echo "gemm report rc=$?"; python - <<P
import json
rows=sorted(json.load(open("$O/gemm_shapes_$TAG.json")), key=lambda r: -r["ms_per_step"])
print(len(rows), "shapes", round(sum(r["ms_per_step"] for r in rows),1), "ms per step")
for r in rows[:14]: print(r["M"], r["N"], r["K"], "act", r["act"], "x", r["launches_per_step"], round(r["ms_per_step"],2), "ms", round(r["TFLOPs"]), "TF", round(r["TFLOPs"]/2500,3))
P
      ;;
    lab) timeout 300 python tools/r05_lab.py "$@" 2>&1 | tail -60 ;;
    pmcflash)   # SQ counter passes (separate --pmc runs, --kernel-trace only) of the flash kernels: pmcflash <knob> <fwd|bwd> <kernel substring...>
      local knob=$1 what=$2; shift 2
      local W=flashonly; [ "$what" == "bwd" ] && W=flashbwdonly
      local P=$GRAFT_REPO_ROOT/$O/pmc_$TAG; mkdir -p $P
      local CMD="python $GRAFT_REPO_ROOT/tools/r05_lab.py $W --knob $knob --B 8"
      ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/sq -o sq -- $CMD > $P/sq.log 2>&1
        timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/sq2 -o sq2 -- $CMD > $P/sq2.log 2>&1 )
      for k in "$@"; do python tools/pmc_sq_summarise.py $P $k | tee -a $O/pmc_flash_$TAG.jsonl | cut -c1-1500; done
      rm -rf $P
      ;;
    pmcsq)   # SQ counter passes (separate --pmc runs, --kernel-trace only) of any command: pmcsq <tag> <kernel substring> <command...>
      local tag=$1 ksub=$2; shift 2
      local P=$GRAFT_REPO_ROOT/$O/pmc_$tag; mkdir -p $P
      ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/sq -o sq -- "$@" > $P/sq.log 2>&1
        timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/sq2 -o sq2 -- "$@" > $P/sq2.log 2>&1
        timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $P/sq3 -o sq3 -- "$@" > $P/sq3.log 2>&1 )
      python tools/pmc_sq_summarise.py $P "$ksub" | tee -a $O/pmc_sq_$tag.jsonl | cut -c1-2500
      tail -3 $P/sq.log; rm -rf $P   # (the command runs from /tmp: give it absolute paths)
      ;;
    py) timeout 1200 python "$@" 2>&1 | tail -60 ;;
    *) echo "unknown verb $verb"; return 2 ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run_verb "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_verb "${args[@]}"
