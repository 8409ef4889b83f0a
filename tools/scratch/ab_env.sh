# same-box A/B of an environment switch: tools/scratch/ab_env.sh VAR v0 v1 [repeats]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
VAR=$1; A=$2; B=$3; R=${4:-2}
for r in $(seq 1 $R); do for v in $A $B; do
  env $VAR=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass > gpurun_out/r06/ab_${VAR}_${v}_$r.json 2> gpurun_out/r06/ab_${VAR}_${v}_$r.err
  python -c "
import json;d=json.loads(open('gpurun_out/r06/ab_${VAR}_${v}_$r.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$VAR=$v run $r', round(d['ms_per_step'],1),'region',round(r.get('region_ms_per_step'),1),'fwd',round(r.get('region_fwd_ms_per_step'),1),'bwd',round(r.get('region_bwd_ms_per_step'),1),'p5 frac',round(r['frac'],3))"
done; done
