#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc_abs; mkdir -p $O
export TMPDIR=/tmp
for k in sva_abs_fwd sva_abs_bwd; do
  ( timeout 120 python tools/bench_hbm_kernels.py --only $k --md $O/hbm_$k.md --json $O/hbm_$k.json > $O/hbm_$k.log 2>&1; tail -3 $O/hbm_$k.md )
  ( timeout 200 bash tools/pmc_hbm.sh $k $k > $O/pmc_$k.log 2>&1 )
  python tools/pmc_hbm_summarise.py gpurun_out/pmc_hbm/$k ${k}_kernel > $O/pmc_$k.json 2>&1; cat $O/pmc_$k.json
  rm -rf gpurun_out/pmc_hbm/$k
done
