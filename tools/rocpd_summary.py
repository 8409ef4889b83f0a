#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (``*_results.db``) into the per-kernel table that
``rocprofv3 --kernel-trace --stats`` prints: calls, total / average / min / max duration, share.

    python tools/rocpd_summary.py gpurun_out/prof/xxx_results.db > profiles/rNN_kernel_stats.md
"""
import sqlite3
import sys


def main(path: str, top: int = 60) -> None:
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"source: {path}\n")
    print(f"total kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx in rows[:top]:
        nm = name if len(name) <= 110 else name[:107] + "..."
        print(f"| `{nm}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
