#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c4; mkdir -p $O
export TMPDIR=/tmp
export CAMBRIAN_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity.jsonl
( timeout 900 python -m pytest tests/test_gemm256_gpu.py tests/test_release_width_gpu.py -m gpu -q -k "rowmap or release_width" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -25 $O/pytest.log
unset CAMBRIAN_PARITY_LOG
( timeout 200 python tools/bench_hbm_kernels.py --md $O/hbm.md --json $O/hbm.json > $O/hbm.log 2>&1; echo "hbm rc=$?" )
for c in "layernorm_bwd [SVA, fp32" "sva_bwd" "sva_fwd"; do
  tag=$(echo "$c" | tr -c 'a-zA-Z0-9' '_' | cut -c1-24)
  timeout 200 bash tools/pmc_hbm.sh "$c" $tag > $O/pmc_$tag.log 2>&1
done
python tools/pmc_hbm_summarise.py gpurun_out/pmc_hbm/layernorm_bwd__SVA__fp32 layernorm_bwd > $O/pmc_ln_bwd.json 2>&1
python tools/pmc_hbm_summarise.py gpurun_out/pmc_hbm/sva_bwd_ sva_bwd > $O/pmc_sva_bwd.json 2>&1
python tools/pmc_hbm_summarise.py gpurun_out/pmc_hbm/sva_fwd_ sva_fwd > $O/pmc_sva_fwd.json 2>&1
cat $O/hbm.md; cat $O/pmc_*.json; cat $O/parity.jsonl
