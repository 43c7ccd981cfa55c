"""Kernel statistics of WHOLE training iterations only, from the per-dispatch listing profiles/prof.sh writes with
PROF_TRACE="<trace.csv> <n>": the process-wide `*_kernel_stats.csv` also counts model construction, the state-dict upload
and the bench loop's own per-step launches, which is why its launches-per-iteration figure is larger than an iteration's.
    python profiles/iteration_stats.py <trace.csv> <out.csv>
One iteration = from one encoder-trunk forward launch to the next; every complete iteration of the listing is used.
Columns: Name, CallsPerIteration, AverageNs, TotalNsPerIteration, Library (0 for at::native / rocclr kernels)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marks = [i for i, r in enumerate(rows) if any(t in r["name"] for t in ("k_trunk4<true>", "k_trunk_bf2<true>", "k_trunk<1, true>",
                                                                       "k_trunk_split<true>"))]
if len(marks) < 3:
    sys.exit("need at least three iterations in the trace")
its = len(marks) - 1
agg = collections.OrderedDict()
for r in rows[marks[0]:marks[-1]]:
    a = agg.setdefault(r["name"], [0, 0])
    a[0] += 1
    a[1] += int(r["duration"])
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "CallsPerIteration", "AverageNs", "TotalNsPerIteration", "Library"])
    for name, (n, t) in out:
        lib = 0 if ("at::native" in name or "rocclr" in name) else 1
        w.writerow([name, round(n / its, 3), round(t / n, 1), round(t / its, 1), lib])
tot = sum(n for _, (n, _) in out) / its
nonlib = sum(n for name, (n, _) in out if "at::native" in name or "rocclr" in name) / its
print(f"{its} iterations: {tot:.1f} launches per iteration, {nonlib:.1f} of them at::native / rocclr, "
      f"{sum(t for _, (_, t) in out) / its / 1e6:.3f} ms of kernel time per iteration -> {sys.argv[2]}")
