"""Kernel summary of a rocprofv3 run (rocpd sqlite output): python tools/rocpd_summary.py results.db [out.csv]
One line per kernel: calls, total ms, average / min / max microseconds, share of the GPU time."""
import csv
import sqlite3
import sys


def summary(db):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    return [(r[0], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total) for r in rows]


if __name__ == "__main__":
    rows = summary(sys.argv[1])
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.4f}", f"{r[3]:.3f}", f"{r[4]:.3f}", f"{r[5]:.3f}", f"{r[6]:.2f}"])
