#!/usr/bin/env python
"""Dump the per-kernel statistics of a rocprofv3 (rocpd sqlite) results database as CSV.

usage: python tools/prof_summary.py gpurun_out/prof_x/<name>_results.db profiles/<name>_kernel_stats.csv
Equivalent to the `--stats` kernel table: name, calls, total (us), average (us), percentage.
"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.2f}"])
    for r in rows[:12]:
        print(f"{r[3]:10.1f} us avg x{r[1]:4d}  {r[4]:5.1f}%  {r[0][:90]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
