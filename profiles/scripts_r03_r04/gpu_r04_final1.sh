#!/bin/bash
# round-4 validation 1: the whole GPU test suite, with the observed parity figures logged
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/r04_parity_observed.jsonl
CAMBRIAN_PARITY_LOG=$PWD/gpurun_out/r04_parity_observed.jsonl timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r04_pytest_gpu_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
