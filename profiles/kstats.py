"""Per-iteration view of a rocprofv3 kernel-stats CSV of a training run (profiles/prof.sh): python profiles/kstats.py <csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
its = [int(r["Calls"]) for r in rows if "k_trunk4<true>" in r["Name"] or "k_trunk_bf2<true>" in r["Name"] or "k_trunk<1, true>" in r["Name"]]
its = its[0] if its else 1
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("iterations", its, "kernel ms/it", round(tot / its / 1e6, 3), "launches/it", round(sum(int(r["Calls"]) for r in rows) / its, 1))
lib = sum(int(r["Calls"]) for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"]) / its
print("at::native + rocclr launches/it", round(lib, 1))
cum = 0.0
for r in rows[:top]:
    nm = r["Name"].split("(")[0][-60:]
    t = int(r["TotalDurationNs"]) / its / 1e3
    cum += t
    print(f"{nm:62s} {int(r['Calls']) / its:6.2f} {float(r['AverageNs']) / 1e3:9.1f} {t:9.1f} {cum:9.1f}")
