#!/bin/bash
# round-4 run 8: finetune stage (tests + NOT_HEADLINE bench lines), tower recompute at the reference batch, comm-only at world 1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_towers_train_gpu.py -m gpu -x -q -k "finetune or recompute or end_to_end" > gpurun_out/r04_lab8_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r04_lab8_pytest.log
X="--no-cpu-baseline --no-ab --no-masked-case --no-gemm-pass --no-calibration"
timeout 900 python bench.py --stage finetune --steps 3 --warmup 1 $X > gpurun_out/r04_bench_finetune_8b.json 2> gpurun_out/r04_bench_finetune_8b.err; echo "finetune rc=$?"
timeout 900 python bench.py --stage finetune --batch 8 --steps 3 --warmup 1 $X > gpurun_out/r04_bench_finetune_8b_b8.json 2> gpurun_out/r04_bench_finetune_8b_b8.err; echo "finetune b8 rc=$?"
timeout 900 python bench.py --stage finetune --zero2 --steps 3 --warmup 1 $X > gpurun_out/r04_bench_finetune_8b_zero2.json 2> gpurun_out/r04_bench_finetune_8b_zero2.err; echo "finetune zero2 rc=$?"
timeout 900 python bench.py --unfreeze-towers --tower-recompute --batch 8 --steps 3 --warmup 1 $X > gpurun_out/r04_bench_unfrozen_towers_b8.json 2> gpurun_out/r04_bench_unfrozen_towers_b8.err; echo "unfrozen b8 rc=$?"
timeout 300 python bench.py --comm-only --stage finetune --steps 3 --warmup 1 > gpurun_out/r04_comm_only_w1.json 2> gpurun_out/r04_comm_only_w1.err; echo "comm-only rc=$?"
python - <<'PY'
import json
for f in ("r04_bench_finetune_8b","r04_bench_finetune_8b_b8","r04_bench_finetune_8b_zero2","r04_bench_unfrozen_towers_b8","r04_comm_only_w1"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        c=d["config"]
        print(f, d["value"], d["unit"], d["ms_per_step"], c.get("images_per_gpu"), c.get("peak_hbm_gb"), c.get("trainable_parameters"), c.get("gradient_buckets"), c.get("loss"))
    except Exception as e:
        print(f, "ERR", e); print(open(f"gpurun_out/{f}.err").read()[-1500:])
PY
