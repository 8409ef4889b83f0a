#!/usr/bin/env python
"""Turn the rocprofv3 --pmc passes of tools/pmc_traffic.sh into the JSON bench.py reads for `roofline.traffic`.

    python tools/pmc_summarise.py gpurun_out/pmc M N K profiles/r02_pmc_gemm_p5.json [source.hip ...]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB-sized units of 64 B x 16 and gfx950
tallies a 128-B read request as 64 B (MI355X_MICROARCH.md, HBM / rocprofv3 section).  The sha256 of the kernel's source
files goes into the JSON; bench.py reports `traffic: null` when the sources have changed since (a stale constant)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def main():
    d, M, N, K, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    srcs = sys.argv[6:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_nt" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        raise SystemExit("no gemm_nt kernel in " + d)
    kernel = max(acc, key=lambda k: len(acc[k]))
    c = {n: sum(v) / len(v) for n, v in acc[kernel].items()}
    fetch, write = c["FETCH_SIZE"] * 2 * 1024, c["WRITE_SIZE"] * 1024
    sha = hashlib.sha256()
    for s in srcs:
        sha.update(open(s, "rb").read())
    res = {"kernel": kernel.split("(")[0], "command": f"tools/pmc_traffic.sh {M} {N} {K} ...; tools/pmc_summarise.py",
           "shape": [M, N, K], "counters_avg_per_launch": c,
           "algorithmic_bytes_per_launch": 2 * (M * K + N * K + M * N),
           "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
           # busy SIMD-cycles of the matrix pipe / (kernel cycles x 1024 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs
           "mfma_pipe_utilisation": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0)
           if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c else None,
           "sources": [os.path.relpath(s) for s in srcs], "sources_sha256": sha.hexdigest() if srcs else None,
           "note": "Separate rocprofv3 --pmc passes (SQ x2, FETCH_SIZE, WRITE_SIZE), --kernel-trace only.  FETCH_SIZE counts "
                   "L2->fabric read requests INCLUDING Infinity-Cache hits: an upper bound on HBM reads."}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in ("kernel", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch",
                                          "mfma_pipe_utilisation")}))


if __name__ == "__main__":
    main()
