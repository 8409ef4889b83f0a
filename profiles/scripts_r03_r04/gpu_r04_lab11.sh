#!/bin/bash
# round-4 lab run 11: what packed f32 vector instructions cost on gfx950 (issue-rate probe), and the two kernels that lean
# on them built without packing: dwconv7x7_col (FILEFLAGS_dwconv=-fno-slp-vectorize) and the p5 epilogue
# (DEFS=-DCMB_P5_SCALAR_EPI) — lab libraries through CAMBRIAN_AMD_LIB
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/r04_lab14.jsonl
timeout 120 tools/valu_probe/valu_probe > gpurun_out/r04_valu_probe.jsonl 2> gpurun_out/r04_valu_probe.err; echo "probe rc=$?"; cat gpurun_out/r04_valu_probe.jsonl
rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
DW=$PWD/cambrian_amd/csrc/libcambrian_amd_lab_dwscalar.so
EP=$PWD/cambrian_amd/csrc/libcambrian_amd_lab_scalarepi.so
for i in 1 2; do
timeout 300 python tools/r04_lab.py --only dw --out gpurun_out/r04_lab14.jsonl --tag pk$i > gpurun_out/r04_lab14.log 2>&1; echo "dw pk rc=$?"
CAMBRIAN_AMD_LIB=$DW timeout 300 python tools/r04_lab.py --only dw --out gpurun_out/r04_lab14.jsonl --tag scalar$i >> gpurun_out/r04_lab14.log 2>&1; echo "dw scalar rc=$?"
timeout 300 python tools/r04_lab.py --only gelu,epi --out gpurun_out/r04_lab14.jsonl --tag pk$i >> gpurun_out/r04_lab14.log 2>&1; echo "epi pk rc=$?"
CAMBRIAN_AMD_LIB=$EP timeout 300 python tools/r04_lab.py --only gelu,epi --out gpurun_out/r04_lab14.jsonl --tag scalar$i >> gpurun_out/r04_lab14.log 2>&1; echo "epi scalar rc=$?"
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r04_lab14.jsonl")]
by={}
for r in rows:
    if "us" in r: by.setdefault((r["kernel"],r["shape"],r.get("variant")),{})[r["tag"]]=r["us"]
for k,v in by.items():
    print(f"{str(k):90s}", "  ".join(f"{t} {v[t]:8.1f}" for t in sorted(v)))
PY
