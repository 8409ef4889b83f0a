#!/bin/bash
# round-4 validation 2: the driver's command (twice), its rocprofv3 kernel table, PMC passes of the dominant kernel on its
# hottest launch shape, the HBM-kernel table (+ PMC bytes of three of its cases), the ZeRO-3 finetune line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
for i in 1 2; do
  timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_cmd_run$i.json 2> gpurun_out/r04_bench_driver_cmd_run$i.err; echo "driver cmd run $i rc=$?"
done
bash tools/gpu_profile_driver_cmd.sh > gpurun_out/r04_profile.log 2>&1; echo "profile rc=$?"
cp gpurun_out/prof_driver/kernel_stats.md gpurun_out/r04b_bench_b24_kernel_stats.md; cp gpurun_out/prof_driver/profiled_run.json gpurun_out/r04b_bench_b24_profiled_run.json
rm -rf gpurun_out/pmc; bash tools/pmc_traffic.sh 98304 6144 1536 2590 1 > gpurun_out/r04_pmc_traffic.log 2>&1; echo "pmc rc=$?"
python tools/pmc_summarise.py gpurun_out/pmc 98304 6144 1536 gpurun_out/r04_pmc_gemm_p5.json cambrian_amd/csrc/gemm_p5.hip cambrian_amd/csrc/gemm_p5_epilogue.inc
timeout 600 python tools/bench_hbm_kernels.py --md gpurun_out/r04_hbm_kernels_table.md --json gpurun_out/r04_hbm_kernels.json > gpurun_out/r04_hbm.log 2>&1; echo "hbm rc=$?"
rm -rf gpurun_out/pmc_hbm
for c in "resample_bilinear [stage 1" "layernorm_fwd [ConvNeXt stage 1" "dwconv7x7 [stage 3" "layernorm_bwd_multi"; do
  tag=$(echo "$c" | tr -c 'a-zA-Z0-9' '_')
  bash tools/pmc_hbm.sh "$c" $tag > /dev/null 2>&1
done
python tools/pmc_hbm_summarise.py "gpurun_out/pmc_hbm/resample_bilinear__stage_1" resample > gpurun_out/r04_pmc_hbm.jsonl
python tools/pmc_hbm_summarise.py "gpurun_out/pmc_hbm/layernorm_fwd__ConvNeXt_stage_1" layernorm_fwd >> gpurun_out/r04_pmc_hbm.jsonl
python tools/pmc_hbm_summarise.py "gpurun_out/pmc_hbm/dwconv7x7__stage_3" dwconv >> gpurun_out/r04_pmc_hbm.jsonl
python tools/pmc_hbm_summarise.py "gpurun_out/pmc_hbm/layernorm_bwd_multi" layernorm_bwd_multi >> gpurun_out/r04_pmc_hbm.jsonl
rm -rf gpurun_out/pmc_hbm/*/fetch gpurun_out/pmc_hbm/*/write gpurun_out/pmc/*/*.db 2>/dev/null
X="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass --no-calibration"
timeout 900 python bench.py --stage finetune --zero3 --steps 3 --warmup 1 $X > gpurun_out/r04_bench_finetune_8b_zero3.json 2> gpurun_out/r04_bench_finetune_8b_zero3.err; echo "finetune zero3 rc=$?"
python - <<'PY'
import json
for f in ("r04_bench_driver_cmd_run1","r04_bench_driver_cmd_run2","r04b_bench_b24_profiled_run","r04_bench_finetune_8b_zero3"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("traffic"), {k:r.get("region",{}).get(k) for k in ("frac","ms_per_step","executed_frac")}, r.get("all_own_gemm",{}).get("frac"), d["config"].get("peak_hbm_gb"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat gpurun_out/r04_pmc_hbm.jsonl; tail -30 gpurun_out/r04_hbm_kernels_table.md | cut -c1-170
