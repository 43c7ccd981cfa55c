#!/usr/bin/env python
"""The roofline table of DESIGN.md, generated from the committed rocprofv3 summaries so that the document cannot drift
from `profiles/`:

    python profiles/make_tables.py [r03]            # prints the markdown block
    python profiles/make_tables.py r03 --write      # rewrites the block between the GENERATED markers in DESIGN.md

Inputs (all under profiles/, tag = round): <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py`, fp32),
<tag>_pmc_summary.csv (separate --pmc passes), <tag>_bf16_kernel_stats.csv / <tag>_split_kernel_stats.csv /
<tag>_bf16_pmc_summary.csv, <tag>_train_kernel_stats.csv / <tag>_train_pmc_summary.csv.  `tests/test_host.py` checks that
the block in DESIGN.md equals what this script prints.

Algorithmic FLOPs per launch are SURVEY.md 8(d)'s per-point figures x the 512 clouds x 1024 points one launch processes at
B=256, N=M=1024 (stated next to each kernel below); peaks are /opt/skills/guides/MI355X_MICROARCH.md's dense figures."""
import csv
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PTS = 256 * 2048  # points per launch at B=256, N=M=1024
FP32_PEAK, BF16_PEAK = 157.3, 2500.0
SPLIT_PEAK = BF16_PEAK / 3

# kernel -> (algorithmic FLOP per point, what it is, peak TFLOP/s or None for HBM / latency kernels)
TRUNK = 2 * (64 * 64 + 64 * 128 + 128 * 512 + 512 * 1024) + 2 * 3 * 64 + 18
STN3D = 2 * (64 * 128 + 128 * 1024) + 2 * 3 * 64
STNKD = 2 * (64 * 64 + 64 * 128 + 128 * 1024) + 2 * 3 * 64 + 18
ROT_L1 = 2 * 2 * (64 * 256 + 256 * 256)          # both heads: layer 0 recomputed + layer 1
ALGO = {
    "k_trunk<1>": (TRUNK, "a3+a5 trunk, conv4 + max", FP32_PEAK),
    "k_stn3d<1>": (STN3D, "a2 STN3d conv stack + max", FP32_PEAK),
    "k_stnkd<1>": (STNKD, "a4 STNkd conv stack + max", FP32_PEAK),
    "k_rot_l1<1>": (ROT_L1, "a9 rot heads: layer 0 (recomputed) + GN0 + GELU + layer 1", FP32_PEAK),
    # round 5: full grids run the encoder with ONE wave per SIMD (256 accumulators per wave)
    "k_trunk4": (TRUNK, "a3+a5 trunk, one wave per SIMD: conv4 as MB8 x NB2 + max", FP32_PEAK),
    "k_stn3d_pair": (STN3D, "a2 STN3d conv stack + max, one wave per SIMD, 128-point pairs", FP32_PEAK),
    "k_stnkd_pair": (STNKD, "a4 STNkd conv stack + max, one wave per SIMD, 128-point pairs", FP32_PEAK),
    "k_stn3d<1, false, true>": (STN3D, "a2 STN3d, one wave per SIMD, 64-point tiles", FP32_PEAK),
    "k_stnkd<1, false, true>": (STNKD, "a4 STNkd, one wave per SIMD, 64-point tiles", FP32_PEAK),
    "k_trunk4<true>": (TRUNK, "training forward: trunk (one wave per SIMD) + activation saves", FP32_PEAK),
    # round 6: the fp32 STN stacks store no activation rows (the backward recomputes them): pair kernels + arg-max epilogue
    "k_stn3d_pair<true>": (STN3D, "training forward: STN3d on 128-point pairs, arg-max epilogue, no row saves", FP32_PEAK),
    "k_stnkd_pair<true>": (STNKD, "training forward: STNkd on 128-point pairs, arg-max epilogue, no row saves", FP32_PEAK),
    "k_trunk_bf2<true>": (TRUNK, "autocast training forward: trunk, bf16 operands, saves", BF16_PEAK),
    "k_stn3d_bf2<true>": (STN3D, "autocast training forward: STN3d + saves", BF16_PEAK),
    "k_stnkd_bf2<true>": (STNKD, "autocast training forward: STNkd + saves", BF16_PEAK),
    "k_rot_l1_bwd_bf<true>": (2 * 2 * 256 * 256, "autocast: rot head layer-1 backward, one head, bf16 rows", BF16_PEAK),
    "k_rot_l0_bwd_bf<true>": (2 * 2 * 64 * 256, "autocast: rot head layer-0 backward, one head, bf16 rows", BF16_PEAK),
    "k_trunk_split<1>": (TRUNK, "trunk, conv3/conv4 as split-bf16 (3 products)", SPLIT_PEAK),
    "k_rot_l1_split": (ROT_L1, "rot heads, split-bf16", SPLIT_PEAK),
    "k_stn3d_split<1>": (STN3D, "STN3d, split-bf16", SPLIT_PEAK),
    "k_stnkd_split<1>": (STNKD, "STNkd, split-bf16", SPLIT_PEAK),
    "k_trunk_bf2": (TRUNK, "trunk, bf16 operands, 128-point pairs", BF16_PEAK),
    "k_trunk_bf": (TRUNK, "trunk, bf16 operands, 64-point tiles", BF16_PEAK),
    "k_rot_l1_bf": (ROT_L1, "rot heads, bf16 operands", BF16_PEAK),
    "k_stn3d_bf": (STN3D, "STN3d, bf16 operands", BF16_PEAK),
    "k_stnkd_bf": (STNKD, "STNkd, bf16 operands", BF16_PEAK),
    "k_stn3d_bf2": (STN3D, "STN3d, bf16 operands, 128-point pairs", BF16_PEAK),
    "k_stnkd_bf2": (STNKD, "STNkd, bf16 operands, 128-point pairs", BF16_PEAK),
    # training (config 3): dense GEMM FLOPs of the op, per row of the [rows, C] activation
    "k_trunk<1, true>": (TRUNK, "training forward: trunk + activation saves", FP32_PEAK),
    "k_stn3d<1, true>": (STN3D, "training forward: STN3d + saves", FP32_PEAK),
    "k_stnkd<1, true>": (STNKD, "training forward: STNkd + saves", FP32_PEAK),
    "k_rot_l1<1, true>": (ROT_L1, "training forward: both rot heads (layer 0 + GN0 + GELU + layer 1) + saves of y0 / a0 / y1", FP32_PEAK),
    "k_rot_l1_bwd": (2 * 2 * 256 * 256, "rot head layer-1 backward (dgrad + wgrad), one head, rows = B*(N+M)", FP32_PEAK),
    "k_rot_l0_bwd": (2 * 2 * 64 * 256, "rot head layer-0 backward (dgrad + wgrad), one head", FP32_PEAK),
    # the row GEMMs serve several layers: FLOPs = what the launches issued on average (PMC: MFMA ops x 512), None here
    "k_gemm_rows<1, 32>": (None, "row GEMM, K = 512 (conv3 dgrad 512 -> 128 on the live rows)", FP32_PEAK),
    "k_gemm_rows<1, 8>": (None, "row GEMM, K = 64 (conv2-class dgrads on the live rows)", FP32_PEAK),
    "k_gemm_rows<1, 16>": (None, "row GEMM, K = 128 (conv2-class dgrads)", FP32_PEAK),
    "k_gemm_tn<2>": (None, "weight gradients, 128 x 128 tiles", FP32_PEAK),
    "k_gemm_tn<1>": (None, "weight gradients, 128 x 64 tiles (K <= 64)", FP32_PEAK),
}


def canon(n):
    """Instances that differ only by a trailing `false` template argument (the non-SAVE forms; the argument was added in
    round 3, so older summaries spell them without it) get one name: k_trunk<1, false> -> k_trunk<1>, k_trunk_bf2<false>
    -> k_trunk_bf2.  `true` (training SAVE) instances keep theirs."""
    n = n.strip()
    n = re.sub(r",\s*false>$", ">", n)
    n = re.sub(r"<false>$", "", n)
    # k_stn*<1, false, false> (two workgroups per CU, the round-3/4 form) -> k_stn*<1>; <1, true, false> -> <1, true>
    n = re.sub(r"^(k_stn[3k]d)<(\d+), (true|false), false>$", lambda m: f"{m.group(1)}<{m.group(2)}" + (", true>" if m.group(3) == "true" else ">"), n)
    return n


def short(name):
    n = name.replace("void ", "").strip()
    return canon(n.split("(")[0].strip())


def stats(path):
    out = {}
    if not os.path.exists(path):
        return out
    with open(path) as f:
        for r in csv.DictReader(f):
            out[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return out


def pmc(path):
    out = {}
    if not os.path.exists(path):
        return out
    with open(path) as f:
        for r in csv.DictReader(f):
            g = lambda k: float(r[k]) if r.get(k) not in (None, "") else None
            gui, busy = g("mean_GRBM_GUI_ACTIVE"), g("mean_SQ_VALU_MFMA_BUSY_CYCLES")
            simd_cycles = gui / 8 * 1024 if gui else None     # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
            fetch, write = g("mean_FETCH_SIZE"), g("mean_WRITE_SIZE")
            mops = g("mean_SQ_INSTS_VALU_MFMA_MOPS_F32")
            out[canon(r["Kernel"])] = dict(
                gflop=mops * 512 / 1e9 if mops else None,
                busy=busy / simd_cycles if busy and simd_cycles else None,
                rd_mb=fetch * 1024 * 2 / 1e6 if fetch is not None else None,   # gfx950 FETCH_SIZE x2 correction (guide)
                wr_mb=write * 1024 / 1e6 if write is not None else None)
    return out


def block(tag, title, stats_file, pmc_file, keys, note):
    st, pm = stats(os.path.join(HERE, stats_file)), pmc(os.path.join(HERE, pmc_file)) if pmc_file else {}
    if not st:
        return [f"*{title}: `profiles/{stats_file}` not collected.*", ""]
    lines = [f"**{title}** (`profiles/{stats_file}`" + (f", `profiles/{pmc_file}`" if pmc_file and pm else "") + f"){note}", "",
             "| kernel | what | GFLOP / launch | rocprof avg µs | TFLOP/s | % of peak | MFMA busy (PMC) | HBM read / write MB (PMC) |",
             "|---|---|---|---|---|---|---|---|"]
    for k in keys:
        if k not in st:
            continue
        calls, us = st[k]
        per_pt, what, peak = ALGO[k]
        p = pm.get(k, {})
        gflop = per_pt * PTS / 1e9 if per_pt else p.get("gflop")
        if gflop is None:
            continue
        tf = gflop / us * 1e3   # GFLOP / µs = PFLOP/s
        busy = f"{100 * p['busy']:.1f} %" if p.get("busy") else "-"
        hbm = f"{p['rd_mb']:.0f} / {p['wr_mb']:.0f}" if p.get("rd_mb") is not None and p.get("wr_mb") is not None else "-"
        lines.append(f"| `{k}` | {what} | {gflop:.1f} | {us:.1f} | {tf:.1f} | {100 * tf / peak:.1f} % of {peak:.0f} | {busy} | {hbm} |")
    rest = sorted(((us * calls, k, calls, us) for k, (calls, us) in st.items() if k not in keys), reverse=True)[:8]
    if rest:
        lines += ["", "Largest other kernels of the same run (no matrix roofline: HBM- or latency-bound): " +
                  "; ".join(f"`{re.sub(r'<.*', '', k)[:40]}` {calls} x {us:.1f} µs" for _, k, calls, us in rest) + "."]
    lines.append("")
    return lines


def _raw_stats(path):
    out = []
    if os.path.exists(path):
        with open(path) as f:
            for r in csv.DictReader(f):
                out.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3))
    return out


def _raw_pmc(path):
    out = []
    if os.path.exists(path):
        with open(path) as f:
            for r in csv.DictReader(f):
                g = lambda k: float(r[k]) if r.get(k) not in (None, "") else 0.0
                out.append(dict(k=canon(r["Kernel"]), n=int(r["Dispatches"]),
                                flop=512.0 * (g("mean_SQ_INSTS_VALU_MFMA_MOPS_F32") + g("mean_SQ_INSTS_VALU_MFMA_MOPS_BF16")),
                                hbm=1024.0 * (2 * g("mean_FETCH_SIZE") + g("mean_WRITE_SIZE"))))
    return out


def budget(title, stats_file, pmc_file, ref, peak, only=None):
    """One line of iteration arithmetic per mode, every figure from the two committed files: MFMA work per iteration
    (PMC MOPS counters x 512 FLOP), HBM bytes per iteration (FETCH_SIZE x 2 + WRITE_SIZE), kernel time per iteration
    (kernel trace), the share of that time spent in launches that carry < 1 % of the FLOPs each (the non-MFMA tail), and the
    fraction of the dense MFMA peak the iteration sustains.  `ref`: the kernel that runs once per iteration; `only`: keep the
    kernels whose name matches (runs that also time another mode)."""
    st, pm = _raw_stats(os.path.join(HERE, stats_file)), _raw_pmc(os.path.join(HERE, pmc_file))
    keep = (lambda k: re.search(only, k) is not None) if only else (lambda k: True)
    refs = (ref,) if isinstance(ref, str) else tuple(ref)   # the once-per-iteration kernel under any of its names
    it_s = sum(c for k, c, _ in st if k in refs)
    it_p = sum(r["n"] for r in pm if r["k"] in refs)
    if not it_s or not it_p:
        return f"| {title} | - | - | - | - | - | - | (`profiles/{stats_file}` / `profiles/{pmc_file}` not collected) |"
    flop = sum(r["n"] * r["flop"] for r in pm if keep(r["k"])) / it_p
    hbm = sum(r["n"] * r["hbm"] for r in pm if keep(r["k"])) / it_p
    big = {r["k"] for r in pm if keep(r["k"]) and r["n"] * r["flop"] / it_p >= 0.01 * flop}
    t_all = sum(c * us for k, c, us in st if keep(k)) / it_s
    t_tail = sum(c * us for k, c, us in st if keep(k) and k not in big) / it_s
    n_all = sum(c for k, c, _ in st if keep(k)) / it_s
    n_tail = sum(c for k, c, _ in st if keep(k) and k not in big) / it_s
    tf = flop / (t_all * 1e-6) / 1e12
    return (f"| {title} | {flop / 1e9:.1f} | {hbm / 1e9:.2f} | {t_all / 1e3:.3f} | {n_all:.0f} | {t_tail / 1e3:.3f} ms in {n_tail:.0f} launches "
            f"({100 * t_tail / t_all:.0f} %) | {tf:.1f} | {tf / peak:.3f} of {peak:.0f} |")


def budgets(tag):
    lines = ["**Iteration arithmetic per mode** (one refine iteration at B=256, N=M=1024; FLOPs and bytes are the PMC passes' counters, "
             "time is the kernel trace's - profiled runs, a few % slower than the un-profiled bench; the tail is every launch that "
             "carries < 1 % of the mode's FLOPs)", "",
             "| mode | MFMA GFLOP / iteration (PMC) | HBM GB / iteration (PMC) | kernel ms / iteration | launches | non-MFMA tail | TFLOP/s | fraction of the dense peak |",
             "|---|---|---|---|---|---|---|---|"]
    no_other = r"^(?!.*(_split|_bf|k_colmax|distribution|reduce_kernel))"
    lines.append(budget("fp32 inference", f"{tag}_kernel_stats.csv", f"{tag}_pmc_summary.csv", ("k_trunk4", "k_trunk<1>"), FP32_PEAK, only=no_other))
    lines.append(budget("bf16 operands", f"{tag}_bf16_kernel_stats.csv", f"{tag}_bf16_pmc_summary.csv", "k_trunk_bf2", BF16_PEAK,
                        only=r"^(?!.*(k_colmax|distribution|reduce_kernel))"))
    lines.append(budget("split mode (issued bf16 products: three per fp32-grade product)", f"{tag}_split_kernel_stats.csv",
                        f"{tag}_split_pmc_summary.csv", "k_trunk_split<1>", BF16_PEAK,
                        only=r"^(?!.*(k_colmax|distribution|reduce_kernel))"))
    lines.append(budget("fp32 training (forward + loss + backward + Ranger)", f"{tag}_train_kernel_stats.csv",
                        f"{tag}_train_pmc_summary.csv", ("k_trunk4<true>", "k_trunk<1, true>"), FP32_PEAK))
    lines.append(budget("autocast training (bf16 operands)", f"{tag}_train_bf16_kernel_stats.csv",
                        f"{tag}_train_bf16_pmc_summary.csv", "k_trunk_bf2<true>", BF16_PEAK))
    lines.append("")
    return lines


def render(tag="r03"):
    out = [f"<!-- BEGIN GENERATED by profiles/make_tables.py {tag} -->"]
    out += block(tag, "fp32 headline path, B=256, N=M=1024, one refine iteration per row",
                 f"{tag}_kernel_stats.csv", f"{tag}_pmc_summary.csv",
                 ["k_trunk4", "k_trunk<1>", "k_stn3d_pair", "k_stn3d<1, false, true>", "k_stn3d<1>", "k_stnkd_pair",
                  "k_stnkd<1, false, true>", "k_stnkd<1>", "k_rot_l1<1>"], "")
    out += block(tag, "split mode (opt-in)", f"{tag}_split_kernel_stats.csv",
                 f"{tag}_split_pmc_summary.csv" if os.path.exists(os.path.join(HERE, f"{tag}_split_pmc_summary.csv")) else f"{tag}_pmc_summary.csv",
                 ["k_trunk_split<1>", "k_stn3d_split<1>", "k_stnkd_split<1>", "k_rot_l1_split"],
                 "; peak = a third of the bf16 dense peak (three products per fp32-grade product)")
    out += block(tag, "bf16 operands (BASELINE config 5 arithmetic)", f"{tag}_bf16_kernel_stats.csv", f"{tag}_bf16_pmc_summary.csv",
                 ["k_trunk_bf2", "k_trunk_bf", "k_stn3d_bf2", "k_stnkd_bf2", "k_stn3d_bf", "k_stnkd_bf", "k_rot_l1_bf"], "")
    out += block(tag, "training iteration (BASELINE config 3), fp32", f"{tag}_train_kernel_stats.csv", f"{tag}_train_pmc_summary.csv",
                 ["k_trunk4<true>", "k_trunk<1, true>", "k_stn3d_pair<true>", "k_stnkd_pair<true>", "k_stn3d<1, true>", "k_stnkd<1, true>", "k_rot_l1<1, true>", "k_rot_l1_bwd", "k_rot_l0_bwd",
                  "k_gemm_rows<1, 32>", "k_gemm_rows<1, 8>", "k_gemm_rows<1, 16>", "k_gemm_tn<2>",
                  "k_gemm_tn<1>"],
                 "; GFLOP = the op's dense GEMM work on B*(N+M) rows")
    out += block(tag, "training iteration under torch.autocast (bf16 operands)", f"{tag}_train_bf16_kernel_stats.csv",
                 f"{tag}_train_bf16_pmc_summary.csv",
                 ["k_trunk_bf2<true>", "k_stn3d_bf2<true>", "k_stnkd_bf2<true>", "k_rot_l1_bwd_bf<true>", "k_rot_l0_bwd_bf<true>"], "")
    out += budgets(tag)
    out.append("<!-- END GENERATED -->")
    return "\n".join(out)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else "r03"
    text = render(tag)
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        doc = open(path).read()
        new = re.sub(r"<!-- BEGIN GENERATED.*?<!-- END GENERATED -->", lambda m: text, doc, flags=re.S)
        if new == doc and text not in doc:
            raise SystemExit("DESIGN.md has no GENERATED block")
        open(path, "w").write(new)
        print("DESIGN.md updated")
    else:
        print(text)


if __name__ == "__main__":
    main()
