"""Parse hipcc's ``-Rpass-analysis=kernel-resource-usage`` remarks (written by ``make -C catre_amd/csrc`` to
``csrc/libcatre_hip.resource_usage.txt`` on every build) into a per-kernel table: registers, scratch, LDS, occupancy.

    python -m catre_amd.resusage [--out profiles/rNN_resource_usage.txt]
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
RAW = os.path.join(HERE, "csrc", "libcatre_hip.resource_usage.txt")

_FIELDS = (("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("ScratchSize [bytes/lane]", "scratch"),
           ("Occupancy [waves/SIMD]", "occupancy"), ("SGPRs Spill", "sgpr_spill"), ("VGPRs Spill", "vgpr_spill"),
           ("LDS Size [bytes/block]", "lds"))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + list(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return [re.sub(r"\(.*$", "", o.replace("void ", "")) for o in out]
    except (OSError, subprocess.CalledProcessError):
        return list(names)


def parse(path=RAW):
    """-> list of dicts (kernel, vgpr, agpr, sgpr, scratch, vgpr_spill, sgpr_spill, lds, occupancy), in source order."""
    rows, cur = [], None
    with open(path) as f:
        for ln in f:
            m = re.search(r"remark: Function Name: (\S+)", ln)
            if m:
                cur = {"mangled": m.group(1)}
                rows.append(cur)
                continue
            if cur is None:
                continue
            for label, key in _FIELDS:
                m = re.search(r"remark:\s+" + re.escape(label) + r": (\d+)", ln)
                if m:
                    cur[key] = int(m.group(1))
    for r, d in zip(rows, demangle([r["mangled"] for r in rows])):
        r["kernel"] = d
    return rows


def table(rows):
    lines = [f"{'kernel':<72} vgpr agpr sgpr scratch vspill sspill     lds occ"]
    for r in rows:
        lines.append(f"{r['kernel'][:72]:<72} {r['vgpr']:>4} {r['agpr']:>4} {r['sgpr']:>4} {r['scratch']:>7} {r['vgpr_spill']:>6} "
                     f"{r['sgpr_spill']:>6} {r['lds']:>7} {r['occupancy']:>3}")
    return "\n".join(lines) + "\n"


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(lib_path):
    """gfx950 code object of ``lib_path`` -> {kernel symbol: [instruction mnemonics]} (llvm-objdump of the offload bundle,
    extracted into a temporary directory).  Used by the ISA lint of tests/test_resources.py."""
    import shutil
    import tempfile

    d = tempfile.mkdtemp(prefix="catre_isa.")
    try:
        so = os.path.join(d, "lib.so")
        shutil.copy(lib_path, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, capture_output=True, cwd=d)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        assert len(co) == 1, os.listdir(d)
        text = subprocess.run([OBJDUMP, "-d", os.path.join(d, co[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", ln)
        if m and cur is not None:
            cur.append(m.group(1))
    return out


def packed_fp32_kernels(lib_path):
    """-> {kernel: count} of kernels whose ISA holds a packed-fp32 VALU op (v_pk_{mul,add,fma}_f32 / v_pk_mov_b32 excluded:
    the move has no arithmetic).  The library is built with the feature off (csrc/Makefile NOPK); see DESIGN.md 6.5."""
    bad = {}
    for k, ins in disassemble(lib_path).items():
        n = sum(1 for i in ins if re.fullmatch(r"v_pk_(mul|add|fma)_f32", i))
        if n:
            bad[k] = n
    return bad


if __name__ == "__main__":
    rows = parse()
    text = table(rows)
    if "--out" in sys.argv:
        with open(sys.argv[sys.argv.index("--out") + 1], "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)
    bad = [r for r in rows if r["scratch"] or r["vgpr_spill"]]
    print(f"{len(rows)} kernels, {len(bad)} with scratch / spills", file=sys.stderr)
