#!/usr/bin/env python
"""GPU idle time between kernels, from the rocpd database of ``rocprofv3 --kernel-trace``.

    python tools/rocpd_timeline.py DB [--skip-frac 0.4] > gpurun_out/timeline.json

Dispatches are sorted by start time; a gap is ``start[i+1] - max(end[..i])`` when positive.  Kernels are classed as
``own`` (this library: cmb_* / anonymous-namespace kernels of cambrian_amd/csrc), ``blas`` (hipBLASLt / Tensile `Cijk_`),
``torch`` (everything else).  Reported per class pair: number of gaps, total idle ms, and the gap-size histogram; plus the
busiest 'previous kernel' names for own->own gaps.  Only the last (1 - skip_frac) of the trace is analysed (steady state).
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict

OWN = re.compile(r"cmb_|gemm_nt|gemm_tn|sva_|layernorm|rmsnorm|rope|dwconv|vit_attn|flash_|resample|patchify|token_mean|"
                 r"embed_splice|copy_rows|transpose|colsum|cast_|act_|splitk|fold_kv|ce_|swiglu|qkv_|quantize|image_resample|"
                 r"add_rmsnorm|gather|scatter")


def klass(name: str) -> str:
    if "Cijk_" in name or "Tensile" in name:
        return "blas"
    if OWN.search(name):
        return "own"
    return "torch"


def main():
    path = sys.argv[1]
    skip = 0.4
    if "--skip-frac" in sys.argv:
        skip = float(sys.argv[sys.argv.index("--skip-frac") + 1])
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    if not cols:   # a view: take the column names from a row
        cur = db.execute("select * from kernels limit 1")
        cols = [d[0] for d in cur.description]
    s_col = "start" if "start" in cols else "start_timestamp"
    e_col = "end" if "end" in cols else "end_timestamp"
    rows = db.execute(f"select name, {s_col}, {e_col} from kernels order by {s_col}").fetchall()
    n0 = int(len(rows) * skip)
    rows = rows[n0:]
    t_first, t_last = rows[0][1], max(r[2] for r in rows)
    busy = 0
    cur_end = rows[0][1]
    gaps = defaultdict(lambda: [0, 0.0])
    hist = defaultdict(lambda: defaultdict(int))
    prev_own = defaultdict(lambda: [0, 0.0])
    prev_name = None
    overlap_ns = 0
    for name, s, e in rows:
        if s > cur_end:
            g = s - cur_end
            if prev_name is not None:
                key = f"{klass(prev_name)}->{klass(name)}"
                gaps[key][0] += 1
                gaps[key][1] += g
                b = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<30us" if g < 30000 else "<100us" if g < 100000 else ">=100us"
                hist[key][b] += 1
                if key == "own->own":
                    short = re.sub(r"<.*", "", prev_name.split("(")[0])[-60:]
                    prev_own[short][0] += 1
                    prev_own[short][1] += g
            busy += e - s
        else:
            ov = min(e, cur_end) - s
            overlap_ns += max(ov, 0)
            busy += max(e - cur_end, 0)
        if e > cur_end:
            cur_end = e
            prev_name = name
    wall = t_last - t_first
    out = {"db": path, "dispatches": len(rows), "wall_ms": wall / 1e6, "busy_ms": busy / 1e6, "idle_ms": (wall - busy) / 1e6,
           "idle_frac": (wall - busy) / wall, "overlap_ms": overlap_ns / 1e6,
           "gaps": {k: {"n": v[0], "idle_ms": v[1] / 1e6, "avg_us": v[1] / v[0] / 1e3, "hist": dict(hist[k])} for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])},
           "own_to_own_by_prev_kernel": {k: {"n": v[0], "idle_ms": v[1] / 1e6, "avg_us": v[1] / v[0] / 1e3}
                                         for k, v in sorted(prev_own.items(), key=lambda kv: -kv[1][1])[:25]}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
