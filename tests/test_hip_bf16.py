"""Reduced-precision path (BASELINE config 5: bf16 GEMM operands, fp32 accumulate / GN statistics / SO(3) update).

Three anchors, tolerances stated here:
  * vs the ORACLE WITH THE SAME OPERAND ROUNDING (``oracle.catre_oracle.operand_rounding("bf16")``), ONE ITERATION AT A
    TIME from the HIP path's own previous estimate: what is left is fp32 re-association plus the rare activation that
    rounds to the neighbouring bf16 value -> 8e-4 abs on R, t, s for zero-centred inputs (measured 3.5e-4 .. 5.4e-4: the
    figure moves with every change of an fp32 summation order, e.g. the ts head's K slices in round 2) and 1.5e-3
    with ZERO_CENTER_INPUT=False, where coordinates are ~10x larger (measured 3.4e-4 .. 5.7e-4 depending on the fp32
    summation order of the FC tails - the rounding flips make this a noisy bound).  (Free-running over K iterations
    the two drift apart to ~1e-3: a 1e-4 change of the fed-back pose moves ~2 % of all activations across a bf16
    rounding boundary - any two bf16 implementations differ by that much, so it is not a useful parity bar.)
  * vs the fp32 REFERENCE goldens: bf16-class, 1e-2 abs (measured <= 6.3e-3);
  * vs the REFERENCE UNDER ITS OWN bf16 AUTOCAST (``tests/golden/amp_bf16_reference.npz``): our deviation from the fp32
    reference must not exceed the reference's own (ours keeps fp32 accumulators and fp32 layer outputs; measured 8x closer).
"""
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN_DIR, golden_names, load_golden, recipe_sd

EMU_TOL = 8e-4
EMU_TOL_UNCENTRED = 1.5e-3
FP32_TOL = 1e-2


def _amp_ref():
    return np.load(os.path.join(GOLDEN_DIR, "amp_bf16_reference.npz"))


def test_oracle_emulation_is_closer_to_fp32_than_reference_autocast():
    """CPU: pins the emulation the GPU test relies on, and the characterisation quoted in DESIGN.md."""
    from oracle import catre_oracle as O

    name = "refine_b2_n1024"
    g, z = load_golden(name), _amp_ref()
    sd = recipe_sd(g["cfg"], g["salt"])
    with O.operand_rounding("bf16"):
        out = O.refine_k(g["batch"], sd, g["cfg"], n_iter=g["K"])
    K = g["K"]
    ours = np.abs(out[f"pose_{K}"].numpy() - g["ref"][f"pose_{K}"]).max()
    theirs = np.abs(z[f"{name}__pose_{K}"] - g["ref"][f"pose_{K}"]).max()
    assert 1e-5 < ours < FP32_TOL and ours < theirs, (ours, theirs)
    # the context manager restores full precision
    out32 = O.refine_k(g["batch"], sd, g["cfg"], n_iter=1)
    assert np.abs(out32["pose_1"].numpy() - g["ref"]["pose_1"]).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names())
def test_bf16_refine_matches_rounding_oracle_and_fp32_reference(name):
    from oracle import catre_oracle as O
    from tests.test_hip_parity import build_model, to_dev

    g = load_golden(name)
    model, sd = build_model(g["cfg"], g["salt"])
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
    out = model.refine(to_dev(g["batch"]), n_iter=g["K"])
    torch.cuda.synchronize()
    for i in range(1, g["K"] + 1):
        step = dict(g["batch"])  # teacher-forced: the oracle starts iteration i from the HIP estimate i-1
        step["obj_pose_est"] = out[f"pose_{i - 1}"].cpu()
        if g["cfg"].MODEL.REFINE_SCLAE:
            step["obj_scale_est"] = out[f"scale_{i - 1}"].cpu()
        with O.operand_rounding("bf16"):
            emu = O.refine_k(step, sd, g["cfg"], n_iter=1)
        for key in ("pose", "scale"):
            got = out[f"{key}_{i}"].cpu().numpy()
            e_emu = np.abs(got - emu[f"{key}_1"].numpy()).max()
            e_ref = np.abs(got - g["ref"][f"{key}_{i}"]).max()
            tol = EMU_TOL if g["cfg"].INPUT.ZERO_CENTER_INPUT else EMU_TOL_UNCENTRED
            assert e_emu <= tol, f"{name} {key}_{i}: {e_emu:.3e} vs the operand-rounding oracle"
            assert e_ref <= FP32_TOL, f"{name} {key}_{i}: {e_ref:.3e} vs the fp32 reference"
    assert np.abs(out["pose_1"].cpu().numpy() - g["ref"]["pose_1"]).max() > 1e-6, "bf16 path did not run"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["refine_b2_n1024", "refine_b1_n2048_k8"])
def test_bf16_no_worse_than_reference_autocast(name):
    from tests.test_hip_parity import build_model, to_dev

    g, z = load_golden(name), _amp_ref()
    model, _ = build_model(g["cfg"], g["salt"])
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
    out = model.refine(to_dev(g["batch"]), n_iter=g["K"])
    for i in (g["K"],):  # the refined estimate; single early iterations of a 1-object batch are too noisy to rank
        for key in (f"pose_{i}", f"scale_{i}"):
            ours = np.abs(out[key].cpu().numpy() - g["ref"][key]).max()
            theirs = np.abs(z[f"{name}__{key}"] - g["ref"][key]).max()
            assert ours <= theirs, f"{name} {key}: ours {ours:.3e} vs reference-autocast {theirs:.3e}"


@pytest.mark.gpu
def test_autocast_selects_bf16_kernels_like_the_reference_amp_switch():
    """engine.py:304 / TEST.AMP_TEST wrap the forward in autocast; that is the switch the drop-in honours."""
    from catre_amd.batching import batch_updater_test
    from tests.test_hip_parity import build_model, to_dev

    g = load_golden("refine_b2_n1024")
    model, _ = build_model(g["cfg"], g["salt"])
    batch = to_dev(g["batch"])
    batch_updater_test(model.cfg, batch)
    args = (batch["x"], batch["tfd_kps"])
    kw = dict(init_pose=batch["obj_pose_est"], init_scale=batch["obj_scale_est"], K_zoom=batch["K"],
              mean_scales=batch["obj_mean_scales"], do_loss=False, cur_iter=1)
    with torch.no_grad():
        p32 = model(*args, **kw)["pose_1"]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            p_amp = model(*args, **kw)["pose_1"]
        model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
        p16 = model(*args, **kw)["pose_1"]
        model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "fp32"
        with torch.autocast("cuda", dtype=torch.bfloat16):
            p_forced32 = model(*args, **kw)["pose_1"]
    assert p_amp.dtype == torch.float32 and torch.equal(p_amp, p16)
    assert torch.equal(p_forced32, p32) and not torch.equal(p16, p32)
    assert np.abs(p32.cpu().numpy() - g["ref"]["pose_1"]).max() < 2e-5
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "fp16x"
    with pytest.raises(ValueError), torch.no_grad():
        model(*args, **kw)


@pytest.mark.gpu
def test_bf16_ragged_and_full_size_properties():
    """Ragged tiles (N, M not multiples of 64) against the rounding oracle, and at the full config-5 shape the
    size-independent properties: finite outputs, R in SO(3), determinism, batch-row independence."""
    from catre_amd import synth
    from oracle import catre_oracle as O
    from tests.test_hip_parity import build_model, to_dev

    g = load_golden("refine_b3_ragged")
    cfg = g["cfg"].__deepcopy__({})
    model, sd = build_model(cfg, 1)
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
    for (B, N, M) in [(1, 1000, 500), (2, 65, 1), (3, 127, 130)]:
        cfg2 = cfg.__deepcopy__({})
        cfg2.INPUT.NUM_PCL, cfg2.INPUT.NUM_KPS = N, M
        cfg2.MODEL.CATRE.ROT_HEAD.INIT_CFG.num_points = N + M
        m2, sd2 = build_model(cfg2, 3)
        m2.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
        b = synth.make_inputs(B, N, M, seed=40 + B)
        out = m2.refine(to_dev(b), n_iter=1)
        with O.operand_rounding("bf16"):
            emu = O.refine_k(b, sd2, cfg2, n_iter=1)
        for key in ("pose_1", "scale_1"):
            assert np.abs(out[key].cpu().numpy() - emu[key].numpy()).max() <= EMU_TOL, (B, N, M, key)

    B, N, M, K = 256, 2048, 1024, 8  # BASELINE.json config 5 at its full per-GPU batch
    cfg5 = cfg.__deepcopy__({})
    cfg5.INPUT.NUM_PCL, cfg5.INPUT.NUM_KPS = N, M
    cfg5.MODEL.CATRE.ROT_HEAD.INIT_CFG.num_points = N + M
    m5, _ = build_model(cfg5, 0)
    m5.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
    b = to_dev(synth.make_inputs(B, N, M, seed=5))
    o1 = m5.refine(b, n_iter=K)
    o2 = m5.refine(b, n_iter=K)
    R = o1[f"pose_{K}"][:, :3, :3]
    assert torch.isfinite(o1[f"pose_{K}"]).all() and torch.isfinite(o1[f"scale_{K}"]).all()
    assert (R @ R.transpose(1, 2) - torch.eye(3, device=R.device)).abs().max() < 1e-4
    assert (torch.linalg.det(R) - 1).abs().max() < 1e-4
    assert torch.equal(o1[f"pose_{K}"], o2[f"pose_{K}"]), "bf16 path must be deterministic"
    half = {k: (v[: B // 2] if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in b.items()}
    o3 = m5.refine(half, n_iter=K)
    assert torch.equal(o3[f"pose_{K}"], o1[f"pose_{K}"][: B // 2]), "objects must be independent of their batch"


@pytest.mark.gpu
@pytest.mark.parametrize("offset", [4.0, -8.0])
def test_bf16_rot_head_with_an_offset_second_layer_matches_the_rounding_oracle(offset):
    """k_rot_l1_bf folds the layer-1 bias into its accumulators and takes the GroupNorm-1 partials of FULL tiles
    in one pass (sum, sum of squares).  With a bias that puts the group means far from zero the plain `sum v^2 - s mean` form
    cancels; the kernel takes both sums around an anchor value of the group instead.  Full tiles (N = M = 128) and ragged ones
    (two-pass form) against the rounding oracle, one iteration, with `layers.3.bias` shifted by `offset` (|mean| / std of a
    group ~ 15-30 with the recipe weights; larger offsets leave bf16's own resolution, not the statistics, as the limit)."""
    from catre_amd import synth
    from oracle import catre_oracle as O
    from tests.test_hip_parity import build_model, to_dev

    g = load_golden("refine_b3_ragged")
    for (B, N, M) in [(3, 128, 128), (2, 100, 70)]:
        cfg = g["cfg"].__deepcopy__({})
        cfg.INPUT.NUM_PCL, cfg.INPUT.NUM_KPS = N, M
        cfg.MODEL.CATRE.ROT_HEAD.INIT_CFG.num_points = N + M
        model, sd = build_model(cfg, 2)
        sd = {k: v.clone() for k, v in sd.items()}
        for h in ("x", "y"):
            sd[f"rot_head.rot_head_{h}.layers.3.bias"] += offset
        model.load_state_dict({k: v.to("cuda:0") for k, v in sd.items()}, strict=True)
        model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
        b = synth.make_inputs(B, N, M, seed=60 + B)
        out = model.refine(to_dev(b), n_iter=1)
        with O.operand_rounding("bf16"):
            emu = O.refine_k(b, sd, cfg, n_iter=1)
        for key in ("pose_1", "scale_1"):
            assert np.abs(out[key].cpu().numpy() - emu[key].numpy()).max() <= EMU_TOL, (offset, B, N, M, key)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,M", [(40, 1024, 1024), (130, 300, 100), (48, 1000, 500)])
def test_bf16_pair_trunk_returns_the_bits_of_the_tile_trunk(B, N, M):
    """Grids of >= 512 tile PAIRS run the bf16 trunk on 128 points per workgroup (`k_trunk_bf2`: half the L2 weight stream per
    MFMA), smaller ones on 64 (`k_trunk_bf`).  Same contraction order per output and exact maxima: an object refined inside the
    big batch must get the very bits it gets in a batch of 3 - incl. clouds with an odd tile count (the last pair holds one
    tile) and ragged last tiles."""
    from catre_amd import synth
    from tests.test_hip_parity import build_model, to_dev

    TN, TM = -(-N // 64), -(-M // 64)
    assert B * ((TN + 1) // 2 + (TM + 1) // 2) >= 512 > 3 * ((TN + 1) // 2 + (TM + 1) // 2)
    g = load_golden("refine_b3_ragged")
    cfg = g["cfg"].__deepcopy__({})
    cfg.INPUT.NUM_PCL, cfg.INPUT.NUM_KPS = N, M
    cfg.MODEL.CATRE.ROT_HEAD.INIT_CFG.num_points = N + M
    model, _ = build_model(cfg, 2)
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
    b = to_dev(synth.make_inputs(B, N, M, seed=60 + B))
    big = model.refine(b, n_iter=2)
    for idx in ([0, 1, 2], [B - 3, B // 2, B - 1]):
        sub = {k: v[idx].contiguous() for k, v in b.items()}
        small = model.refine(sub, n_iter=2)
        for key in ("pose_1", "scale_1", "pose_2", "scale_2"):
            assert torch.equal(small[key], big[key][idx]), (key, idx)
