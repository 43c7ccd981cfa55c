#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`--kernel-trace --stats`, ROCm 7.2 default output) as CSV.

    python profiles/export_rocprof.py gpurun_out/prof1/r01_results.db profiles/r01_kernel_stats.csv

Columns follow rocprofv3's own kernel_stats.csv: Name, Calls, TotalDurationNs, AverageNs, Percentage
(+ MinNs/MaxNs computed from the dispatch table).
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
        "group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for name, calls, tot, avg, mn, mx in rows:
            w.writerow([name, calls, int(tot), round(avg, 1), int(mn), int(mx), round(100.0 * tot / total, 3)])
    print(f"wrote {out}: {len(rows)} kernels, {total / 1e6:.3f} ms of kernel time")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
