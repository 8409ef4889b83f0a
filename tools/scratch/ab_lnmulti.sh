#!/bin/bash
# (1) GPU tests of the knobs touched this session; (2) sva_abs_fwd: product library vs the lab build with 2 waves per SIMD;
# (3) same-box A/B of 5 vs 4 layers per layernorm_bwd_multi launch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "layernorm_bwd_multi or colsum or layernorm_bwd_rows or knob" 2>&1 | tail -3
for lib in "" cambrian_amd/csrc/libcambrian_amd_lab_wpe.so; do
  for k in sva_abs_fwd; do
    CAMBRIAN_AMD_LIB=$lib timeout 300 python tools/bench_hbm_kernels.py --only $k --iters 30 --json $O/hbm_wpe_$(basename "${lib:-product}").json 2>&1 | tail -2 | cut -c1-300
  done
done
for rep in 1 2; do
  for v in old new; do
    K=""; F=5; [ $v == old ] && K="4=4" && F=4
    CAMBRIAN_AMD_KNOBS=$K CAMBRIAN_AMD_DEFER_LN_FLUSH=$F timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass > $O/ab_lnchunk_${v}_$rep.json 2> $O/ab_lnchunk_${v}_$rep.err
    python - <<P
import json
d=json.loads(open("$O/ab_lnchunk_${v}_$rep.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$v $rep", round(d["ms_per_step"],1), "region", round(r["region_ms_per_step"],2), "fwd", round(r["region_fwd_ms_per_step"],2), "bwd", round(r["region_bwd_ms_per_step"],2), "peak GB", d["config"].get("peak_hbm_gb"))
P
  done
done
