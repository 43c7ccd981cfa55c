"""No kernel of the product library may use scratch memory or spill vector registers (four kernels
did while DESIGN.md said none).  `make -C catre_amd/csrc` records hipcc's -Rpass-analysis=kernel-resource-usage remarks
next to the library on every build; this test parses them."""
import os

from catre_amd import hip, resusage


def _rows():
    if not os.path.exists(resusage.RAW) or not os.path.exists(hip.LIB_PATH) or \
            os.path.getmtime(resusage.RAW) + 120 < os.path.getmtime(hip.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return resusage.parse()


def test_no_kernel_uses_scratch_or_spills_vgprs():
    rows = _rows()
    assert len(rows) >= 150, len(rows)
    bad = [(r["kernel"], r["scratch"], r["vgpr_spill"]) for r in rows if r["scratch"] or r["vgpr_spill"]]
    assert not bad, f"kernels with scratch / spilled VGPRs: {bad}"
    assert all(r["vgpr"] + r["agpr"] <= 512 for r in rows)


def test_no_packed_fp32_valu_op_ships():
    """v_pk_{mul,add,fma}_f32 returned wrong low halves next to co-resident bf16-MFMA waves.  The
    library is compiled with the target feature off (csrc/Makefile NOPK); this disassembles the shipped gfx950 code object
    and fails on any such instruction (115 of 195 kernels carried the pattern)."""
    _rows()
    if not os.path.exists(resusage.OBJDUMP):
        import pytest

        pytest.skip("llvm-objdump not in this image")
    isa = resusage.disassemble(hip.LIB_PATH)
    assert len(isa) >= 150, len(isa)
    assert any("mfma" in i for ins in isa.values() for i in ins)          # it is the real code object
    bad = resusage.packed_fp32_kernels(hip.LIB_PATH)
    assert not bad, f"packed fp32 VALU ops in {len(bad)} kernels: {sorted(bad.items())[:5]}"


def test_headline_kernels_keep_their_occupancy_shape():
    """The workgroup shapes DESIGN.md section 3 relies on: LDS bytes and registers allow the stated workgroups per CU."""
    by = {r["kernel"]: r for r in _rows()}
    t = by["k_trunk<1, false>"]
    assert t["lds"] == 160 * 1024 and t["vgpr"] <= 256           # one 512-thread workgroup per CU
    for k in ("k_stn3d<1, false, false>", "k_stnkd<1, false, false>", "k_stn3d<1, true, false>", "k_stnkd<1, true, false>",
              "k_rot_l1<1, false>", "k_rot_l1<1, true>", "k_rot_l1_split<false>", "k_rot_l1_split<true>"):
        assert by[k]["lds"] <= 80 * 1024 and by[k]["vgpr"] <= 256, k  # two 256-thread workgroups per CU
    # round 5, full grids: ONE 256-thread workgroup per CU, one wave per SIMD with the whole register file - 256 accumulators
    # (AGPRs) next to <= 256 VGPRs, nothing spilled (the scratch / spill lint above covers them too)
    for k in ("k_trunk4<false>", "k_trunk4<true>", "k_stn3d_pair<false>", "k_stnkd_pair<false>", "k_stn3d_pair<true>",
              "k_stnkd_pair<true>", "k_stn3d<1, false, true>", "k_stnkd<1, false, true>", "k_rot_l1w"):
        assert by[k]["vgpr"] <= 256 and by[k]["lds"] <= 160 * 1024, k
    assert by["k_trunk4<false>"]["lds"] == 160 * 1024


def test_committed_table_lists_every_kernel_of_the_build():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "r06_resource_usage.txt")
    text = open(path).read()
    names = {ln[:72].strip() for ln in text.splitlines()[1:]}
    missing = [r["kernel"][:72] for r in _rows() if r["kernel"][:72].strip() not in names]
    assert not missing, f"profiles/r06_resource_usage.txt is stale (python -m catre_amd.resusage --out ...): {missing[:5]}"
