#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c3; mkdir -p $O
export TMPDIR=/tmp
export CAMBRIAN_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity.jsonl
( timeout 900 python -m pytest tests/test_gemm256_gpu.py tests/test_full_depth_gpu.py tests/test_release_width_gpu.py tests/test_zz_bench_multirank_gpu.py -m gpu -q -k "tail_split or full_depth or release_width or zz or two_ranks or plain_launch" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -25 $O/pytest.log
unset CAMBRIAN_PARITY_LOG
( timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --gemm-report $O/gemm_shapes.json > $O/bench_split.json 2> $O/bench_split.err; echo "bench split rc=$?" )
( CMB_GEMM_NO_TAIL_SPLIT=1 timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass > $O/bench_nosplit.json 2> $O/bench_nosplit.err; echo "bench nosplit rc=$?" )
( timeout 200 python tools/bench_hbm_kernels.py --md $O/hbm.md --json $O/hbm.json > $O/hbm.log 2>&1; echo "hbm rc=$?" )
for c in "layernorm_bwd [SVA, fp32" "sva_bwd" "layernorm_fwd [ConvNeXt stage 3"; do
  tag=$(echo "$c" | tr -c 'a-zA-Z0-9' '_' | cut -c1-24)
  timeout 200 bash tools/pmc_hbm.sh "$c" $tag > $O/pmc_$tag.log 2>&1
done
python tools/pmc_hbm_summarise.py gpurun_out/pmc_hbm/layernorm_bwd__SVA__fp32 layernorm_bwd > $O/pmc_ln_bwd.json 2>&1
python tools/pmc_hbm_summarise.py gpurun_out/pmc_hbm/sva_bwd_________________ sva_bwd > $O/pmc_sva_bwd.json 2>&1
python tools/pmc_hbm_summarise.py gpurun_out/pmc_hbm/layernorm_fwd__ConvNeXt_ layernorm_fwd > $O/pmc_ln_fwd.json 2>&1
ls gpurun_out/pmc_hbm
( timeout 300 bash tools/pmc_traffic.sh 65536 6144 1536 2590 1 > $O/pmc_p5.log 2>&1; python tools/pmc_summarise.py gpurun_out/pmc 65536 6144 1536 $O/pmc_gemm_p5.json cambrian_amd/csrc/gemm_p5.hip cambrian_amd/csrc/gemm_p5_epilogue.inc >> $O/pmc_p5.log 2>&1; rm -rf gpurun_out/pmc_p5; mv gpurun_out/pmc gpurun_out/pmc_p5 )
( timeout 300 bash tools/pmc_traffic.sh 65536 6144 1536 2560 1 > $O/pmc_w8.log 2>&1; python tools/pmc_summarise.py gpurun_out/pmc 65536 6144 1536 $O/pmc_gemm_w8.json cambrian_amd/csrc/gemm256.hip >> $O/pmc_w8.log 2>&1; rm -rf gpurun_out/pmc_w8; mv gpurun_out/pmc gpurun_out/pmc_w8 )
( timeout 300 python3 bench.py --preset 13b --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass > $O/bench_13b.json 2> $O/bench_13b.err; echo "13b rc=$?" )
( timeout 400 python3 bench.py --preset 34b --batch 4 --zero3 --steps 3 --warmup 1 --no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass > $O/bench_34b.json 2> $O/bench_34b.err; echo "34b rc=$?" )
python - <<'P'
import json
for f in ("bench_split","bench_nosplit","bench_13b","bench_34b"):
    try:
        d=json.load(open(f"gpurun_out/c3/{f}.json"))
        r=d.get("roofline",{})
        print(f, round(d["ms_per_step"],1), round(d["value"],3), "frac",round(r.get("frac",0),3), "region",round(r.get("region",{}).get("ms_per_step",0),1), round(r.get("region",{}).get("frac",0),3), "gemm", r.get("all_own_gemm",{}).get("ms_per_step"), "mem", d["config"].get("peak_hbm_gb"))
    except Exception as e: print(f, repr(e)[:200])
P
cat $O/hbm.md; cat $O/pmc_*.json
