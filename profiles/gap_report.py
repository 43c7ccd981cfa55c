"""Where the GPU idles inside one training iteration: from the per-dispatch listing profiles/prof.sh writes with
PROF_TRACE="<trace.csv> <n>" (name, start, duration in ns, launch order).
    python profiles/gap_report.py <trace.csv> [top]
One iteration = from one encoder-trunk forward launch to the next.  Prints span, summed kernel time, idle time (gaps
between the end of everything launched so far and the next start) and the largest gaps with the kernels either side."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
d = [(r["name"].split("(")[0][-48:], int(r["start"]), int(r["duration"])) for r in rows]
marks = [i for i, (n, _, _) in enumerate(d) if "k_trunk4<true>" in n or "k_trunk_bf2<true>" in n or "k_trunk<1, true>" in n
         or "k_trunk_split<true>" in n]
if len(marks) < 3:
    sys.exit("need at least three iterations in the trace")
for a, b in zip(marks[-3:-1], marks[-2:]):
    it = d[a:b]
    t0 = it[0][1]
    span = d[b][1] - t0
    busy = sum(x[2] for x in it)
    end = t0
    idle = 0
    gaps = []
    for k, (n, s, du) in enumerate(it):
        if s > end:
            idle += s - end
            gaps.append((s - end, it[k - 1][0] if k else "-", n, it[k - 1][2] if k else 0, du))
        end = max(end, s + du)
    # the gap up to the next iteration's first launch
    if d[b][1] > end:
        idle += d[b][1] - end
        gaps.append((d[b][1] - end, it[-1][0], d[b][0], it[-1][2], d[b][2]))
    print(f"iteration of {len(it)} launches: span {span / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle {idle / 1e3:.1f} us "
          f"({len(gaps)} gaps, median {sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3:.2f} us)")
gaps.sort(reverse=True)
print("largest gaps of the last iteration: gap us | after (its us) -> before (its us)")
for g, a, b, da, db in gaps[:top]:
    print(f"{g / 1e3:8.2f} | {a:48s} ({da / 1e3:7.1f}) -> {b:48s} ({db / 1e3:7.1f})")
hist = [0] * 6
for g in gaps:
    hist[min(5, int(g[0] / 1e3 // 2))] += g[0]
print("idle by gap size [0-2, 2-4, 4-6, 6-8, 8-10, >10 us]:", [round(h / 1e3, 1) for h in hist])
