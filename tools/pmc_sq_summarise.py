#!/usr/bin/env python
"""Average SQ counters per launch of one kernel from the passes of tools/pmc_sq.sh -> one JSON line.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md)."""
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
    res = {"kernel_substring": sub, "launches": {k: len(v) for k, v in acc.items()}, "avg": c}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
            if k in c:
                res[k + "/SQ_WAVE_CYCLES"] = c[k] / wc
    print(json.dumps(res))


if __name__ == "__main__":
    main()
