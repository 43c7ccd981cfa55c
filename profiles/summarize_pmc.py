#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV passes (one directory per pass) per kernel: mean counter value per dispatch.

    python profiles/summarize_pmc.py gpurun_out/pmc profiles/r02_pmc_summary.csv r02

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced streaming reads, so the read side is
doubled; WRITE_SIZE is uncalibrated (reported as is).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(root, out, tag="r02"):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                acc[name]["_dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    counters = sorted({c for k in acc.values() for c in k if not c.startswith("_")})
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Dispatches"] + [f"mean_{c}" for c in counters] + ["mean_profiled_duration_us"])
        for name, d in sorted(acc.items(), key=lambda kv: -sum(kv[1]["_dur_ns"])):
            n = max(len(v) for k, v in d.items() if not k.startswith("_"))
            w.writerow([name, n] + [round(sum(d[c]) / len(d[c]), 3) if d.get(c) else "" for c in counters]
                       + [round(sum(d["_dur_ns"]) / len(d["_dur_ns"]) / 1e3, 2)])
    t = acc.get("k_trunk4<false>") or acc.get("k_trunk4") or acc.get("k_trunk<1, false>") or acc.get("k_trunk<1>") or acc.get("k_trunk")
    if t and t.get("FETCH_SIZE") and t.get("WRITE_SIZE"):
        fetch = sum(t["FETCH_SIZE"]) / len(t["FETCH_SIZE"]) * 1024
        write = sum(t["WRITE_SIZE"]) / len(t["WRITE_SIZE"]) * 1024
        js = {
            "kernel": "k_trunk4<false>" if (acc.get("k_trunk4<false>") or acc.get("k_trunk4")) else "k_trunk<1, false>", "collected_with": "profiles/pmc.sh (separate --pmc passes) on the build of this commit",
            "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes_raw": write,
            "hbm_bytes_per_launch": 2 * fetch + write,
            "note": "read side doubled per MI355X_MICROARCH.md gfx950 FETCH_SIZE correction; WRITE_SIZE uncalibrated",
        }
        with open(os.path.join(os.path.dirname(out), f"{tag}_trunk_hbm_bytes.json"), "w") as f:
            json.dump(js, f, indent=1)
        print(js)
    print(f"wrote {out}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
