#!/bin/bash
# round-4 lab run 13: the dK/dV attention kernel with the mask as one wave-uniform block and batched softmax chains, against
# the previous library (cambrian_amd/csrc/libcambrian_amd_lab_prev.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/r04_lab19.jsonl
OLD=$PWD/cambrian_amd/csrc/libcambrian_amd_lab_prev.so
timeout 900 python -m pytest tests/test_flash_bwd_gpu.py -m gpu -x -q > gpurun_out/r04_lab19_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_lab19_pytest.log
for i in 1 2; do
timeout 300 python tools/r04_lab.py --only flash --out gpurun_out/r04_lab19.jsonl --tag new$i > gpurun_out/r04_lab19.log 2>&1; echo "lab new rc=$?"
CAMBRIAN_AMD_LIB=$OLD timeout 300 python tools/r04_lab.py --only flash --out gpurun_out/r04_lab19.jsonl --tag old$i >> gpurun_out/r04_lab19.log 2>&1; echo "lab old rc=$?"
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r04_lab19.jsonl")]
for r in rows: print({k:r[k] for k in r if k in ("kernel","shape","us","tag","fwd_us","bwd_us","variant")})
PY
