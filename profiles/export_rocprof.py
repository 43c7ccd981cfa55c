#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`--kernel-trace --stats`, ROCm 7.2 default output) as CSV.

    python profiles/export_rocprof.py gpurun_out/prof1/r01_results.db profiles/r01_kernel_stats.csv

Columns follow rocprofv3's own kernel_stats.csv: Name, Calls, TotalDurationNs, AverageNs, Percentage
(+ MinNs/MaxNs computed from the dispatch table).  Optional third / fourth argument: a CSV for the per-dispatch listing
of the last N dispatches (default 2000) in launch order.
"""
import csv
import sqlite3
import sys


def main(db, out, trace=None, last=2000):
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
    if trace:
        # per-dispatch listing (launch order) of the LAST `last` dispatches: which call of a kernel costs what
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        want = [k for k in ("name", "start", "duration", "grid_x", "grid_y", "grid_z", "workgroup_x", "lds_size",
                            "scratch_size") if k in cols]
        disp = list(c.execute(f"select {', '.join(want)} from kernels order by start"))[-last:]
        with open(trace, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(want)
            w.writerows(disp)
        print(f"wrote {trace}: {len(disp)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *(sys.argv[3:4]), *(int(a) for a in sys.argv[4:5]))
