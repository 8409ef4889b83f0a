#!/bin/bash
# round-4 validation 3: same-box A/B of the library before / after the no-contraction epilogues, the PMC bytes of four
# HBM-table cases, the ZeRO-3 finetune line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
X="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass"
PRE=$PWD/cambrian_amd/csrc/libcambrian_amd_precontract.so
for i in 1 2; do
  timeout 300 python bench.py --steps 6 --warmup 2 $X > gpurun_out/r04_ab11_cur$i.json 2> gpurun_out/r04_ab11_cur$i.err; echo "cur rc=$?"
  CAMBRIAN_AMD_LIB=$PRE timeout 300 python bench.py --steps 6 --warmup 2 $X > gpurun_out/r04_ab11_pre$i.json 2> gpurun_out/r04_ab11_pre$i.err; echo "pre rc=$?"
done
rm -rf gpurun_out/pmc_hbm
: > gpurun_out/r04_pmc_hbm.jsonl
while IFS='|' read -r c k; do
  tag=$(printf '%s' "$c" | tr -c 'a-zA-Z0-9' '_')
  bash tools/pmc_hbm.sh "$c" $tag > gpurun_out/pmc_hbm_$tag.log 2>&1
  python tools/pmc_hbm_summarise.py "gpurun_out/pmc_hbm/$tag" $k >> gpurun_out/r04_pmc_hbm.jsonl
done <<'CASES'
resample_bilinear [stage 1|resample
layernorm_fwd [ConvNeXt stage 1|layernorm_fwd
dwconv7x7 [stage 3|dwconv
layernorm_bwd_multi|layernorm_bwd_multi
CASES
rm -rf gpurun_out/pmc_hbm/*/fetch gpurun_out/pmc_hbm/*/write 2>/dev/null
Y="$X --no-calibration"
timeout 900 python bench.py --stage finetune --zero3 --steps 3 --warmup 1 $Y > gpurun_out/r04_bench_finetune_8b_zero3.json 2> gpurun_out/r04_bench_finetune_8b_zero3.err; echo "finetune zero3 rc=$?"
python - <<'PY'
import json
for f in ("r04_ab11_cur1","r04_ab11_pre1","r04_ab11_cur2","r04_ab11_pre2","r04_bench_finetune_8b_zero3"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f, d["value"], d["ms_per_step"], r.get("frac"), {k:r.get("region",{}).get(k) for k in ("frac","ms_per_step","executed_frac")}, r.get("all_own_gemm",{}).get("frac"), d["config"].get("peak_hbm_gb"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat gpurun_out/r04_pmc_hbm.jsonl; tail -5 gpurun_out/r04_bench_finetune_8b_zero3.err
