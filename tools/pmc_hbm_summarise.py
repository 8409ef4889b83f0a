#!/usr/bin/env python
"""Average FETCH_SIZE / WRITE_SIZE per launch of one kernel from the passes of tools/pmc_hbm.sh.

    python tools/pmc_hbm_summarise.py gpurun_out/pmc_hbm/<tag> <kernel-name substring>   -> one JSON line

HBM bytes per launch = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (MI355X_MICROARCH.md §HBM: rocprofv3 on gfx950 tallies a
128-byte read request as 64 B; WRITE_SIZE is taken as reported)."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d, sub = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    n = {k: len(v) for k, v in acc.items()}
    res = {"kernel_substring": sub, "launches_counted": n, "FETCH_SIZE_avg": c.get("FETCH_SIZE"), "WRITE_SIZE_avg": c.get("WRITE_SIZE"),
           "fetch_bytes_corrected": None if "FETCH_SIZE" not in c else c["FETCH_SIZE"] * 2 * 1024,
           "write_bytes": None if "WRITE_SIZE" not in c else c["WRITE_SIZE"] * 1024}
    if res["fetch_bytes_corrected"] is not None and res["write_bytes"] is not None:
        res["hbm_bytes_per_launch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
