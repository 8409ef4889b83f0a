cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for ts in 0 1 abc0 aab0 ab00 a0b0 0; do CAMBRIAN_AMD_TOWER_STREAMS=$ts timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass > gpurun_out/r06/ts_$ts.json 2> gpurun_out/r06/ts_$ts.err; python -c "
import json;d=json.loads(open('gpurun_out/r06/ts_$ts.json').read().strip().splitlines()[-1]);r=d['roofline'];print('tower_streams=$ts',round(d['ms_per_step'],1),'region',round(r.get('region_ms_per_step'),1),'fwd',round(r.get('region_fwd_ms_per_step'),1),'p5 frac',round(r['frac'],3))"; done
