"""Pin the oracle (oracle/catre_oracle.py) against outputs of the reference itself.

The goldens in tests/golden/ were produced by oracle/make_golden.py, which imports the
unmodified reference from /root/reference (build container only).  Tolerance: both sides are
fp32 torch-CPU arithmetic of the same op sequence, so they agree to a few ulp; 2e-6 abs.
"""
import numpy as np
import pytest
import torch

from oracle import catre_oracle as O
from tests.util import golden_names, load_golden, recipe_sd

TOL = 2e-6


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_outputs(name):
    g = load_golden(name)
    sd = recipe_sd(g["cfg"], g["salt"])
    torch.set_num_threads(4)
    with torch.no_grad():
        out = O.refine_k(g["batch"], sd, g["cfg"], n_iter=g["K"], detail_iter=1)
    for i in range(g["K"] + 1):
        np.testing.assert_allclose(out[f"pose_{i}"].numpy(), g["ref"][f"pose_{i}"], atol=TOL, rtol=0, err_msg=f"pose_{i}")
        np.testing.assert_allclose(out[f"scale_{i}"].numpy(), g["ref"][f"scale_{i}"], atol=TOL, rtol=0, err_msg=f"scale_{i}")
    d, ref = out["detail"], g["ref"]
    pairs = {
        "stage_trans_x": d["trans_x"], "stage_trans_k": d["trans_k"],
        "stage_transfeat_x": d["transfeat_x"], "stage_transfeat_k": d["transfeat_k"],
        "stage_g_x": d["g_x"], "stage_g_k": d["g_k"],
        "stage_pointfeat_x": d["pointfeat_x"][:, :, :64], "stage_pointfeat_k": d["pointfeat_k"][:, :, :64],
        "stage_pointfeat_max_x": d["pointfeat_x"].max(2)[0],
        "stage_rot_deltas": d["rot_deltas"], "stage_trans_deltas": d["trans_deltas"],
        "stage_scale_deltas": d["scale_deltas"],
    }
    for k, v in pairs.items():
        if v is None:  # no feature transform in this config
            assert k not in ref
            continue
        np.testing.assert_allclose(v.numpy(), ref[k], atol=5e-6, rtol=1e-5, err_msg=k)


def test_oracle_identities():
    """Known-answer identities the reference's own smoke prints aim at."""
    g = torch.Generator().manual_seed(0)
    q = torch.randn(5, 4, generator=g)
    R = O.quat2mat_torch(q)
    # rot6d of a rotation's first two columns returns that rotation (rot_reps.py:640-645 intent)
    d6 = torch.cat([R[:, :, 0], R[:, :, 1]], 1)
    np.testing.assert_allclose(O.rot6d_to_mat_batch(d6).numpy(), R.numpy(), atol=1e-6)
    # dR = I, vz = 1, vxy = 0, ds = 0 leaves the pose unchanged (pose_scale_from_delta_init.py:57,73-74,79-80,93)
    t = torch.tensor([[0.1, -0.2, 1.0]]).repeat(5, 1)
    s = torch.full((5, 3), 0.2)
    K = torch.eye(3).repeat(5, 1, 1) * 500
    R2, t2, s2 = O.pose_scale_from_delta_init(
        torch.eye(3).repeat(5, 1, 1), torch.tensor([[0.0, 0.0, 1.0]]).repeat(5, 1), torch.zeros(5, 3), R, t, s,
        Ks=K, K_aware=True, delta_T_space="image", scale_type="iter_add")
    np.testing.assert_allclose(R2.numpy(), R.numpy(), atol=1e-7)
    np.testing.assert_allclose(t2.numpy(), t.numpy(), atol=1e-7)
    np.testing.assert_allclose(s2.numpy(), s.numpy(), atol=0)


def test_oracle_maxpool_permutation_invariant():
    g = load_golden("refine_b2_small")
    sd = recipe_sd(g["cfg"], g["salt"])
    b = g["batch"]
    perm = torch.randperm(b["pcl"].shape[1], generator=torch.Generator().manual_seed(1))
    x, k = O.pose_apply(b["pcl"], b["obj_kps"], b["obj_pose_est"], b["obj_scale_est"])
    with torch.no_grad():
        _, d1 = O.pointnet_feat(x, sd, detail=True)
        _, d2 = O.pointnet_feat(x[:, :, perm], sd, detail=True)
    np.testing.assert_allclose(d1["g"].numpy(), d2["g"].numpy(), atol=1e-6)
    np.testing.assert_allclose(d1["trans"].numpy(), d2["trans"].numpy(), atol=1e-6)


def test_oracle_training_losses_and_grads_match_reference():
    """Oracle forward + loss + torch autograd vs the reference's own training iteration (do_loss=True, backward):
    the six loss terms, and per parameter the gradient norm and its first 64 entries."""
    from tests.util import load_train_golden

    g = load_train_golden("train_b4")
    cfg, b = g["cfg"], g["batch"]
    sd = {k: v.clone().requires_grad_(True) for k, v in recipe_sd(cfg, g["salt"]).items()}
    x, tfd = O.pose_apply(b["pcl"], b["obj_kps"], b["obj_pose_est"], b["obj_scale_est"])
    pose, scale = O.model_forward(x, tfd, b["obj_pose_est"], b["obj_scale_est"], sd, cfg, K_zoom=b["K"],
                                  mean_scales=b["obj_mean_scales"])
    ld = O.catre_loss(pose[:, :, :3], pose[:, :, 3], scale, b["gt_rot"], b["gt_trans"], b["gt_scale"], b["obj_kps"],
                      g["sym_info"], cfg.MODEL.CATRE.LOSS_CFG)
    ref = g["ref"]
    assert set(ld) == {k[6:] for k in ref if k.startswith("loss__")}
    for k, v in ld.items():
        np.testing.assert_allclose(v.item(), ref[f"loss__{k}"][0], rtol=2e-5, atol=1e-7, err_msg=k)
    sum(ld.values()).backward()
    n_none = 0
    for k, p in sd.items():
        if f"gradnone__{k}" in ref:
            assert p.grad is None
            n_none += 1
            continue
        nrm = float(ref[f"gradnorm__{k}"][0])
        np.testing.assert_allclose(float(p.grad.norm()), nrm, rtol=2e-4, atol=1e-9, err_msg=k)
        np.testing.assert_allclose(p.grad.reshape(-1)[:64].numpy(), ref[f"gradhead__{k}"], atol=2e-4 * nrm + 1e-9, rtol=0,
                                   err_msg=k)
    assert n_none == 6


def _oracle_train_iteration(g, mode, dtype=torch.float32):
    """One training iteration through the oracle under ``operand_rounding(mode)`` -> (losses, gradients)."""
    cfg, b = g["cfg"], g["batch"]
    sd = {k: v.clone().to(dtype).requires_grad_(True) for k, v in recipe_sd(cfg, g["salt"]).items()}
    bb = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in b.items()}
    with O.operand_rounding(mode):
        x, tfd = O.pose_apply(bb["pcl"], bb["obj_kps"], bb["obj_pose_est"], bb["obj_scale_est"])
        pose, scale = O.model_forward(x, tfd, bb["obj_pose_est"], bb["obj_scale_est"], sd, cfg, K_zoom=bb["K"],
                                      mean_scales=bb["obj_mean_scales"])
    ld = O.catre_loss(pose[:, :, :3], pose[:, :, 3], scale, bb["gt_rot"], bb["gt_trans"], bb["gt_scale"], bb["obj_kps"],
                      g["sym_info"], cfg.MODEL.CATRE.LOSS_CFG)
    sum(ld.values()).backward()
    return {k: float(v.detach()) for k, v in ld.items()}, {k: p.grad for k, p in sd.items() if p.grad is not None}


def amp_fixture(name="train_b4"):
    """tests/golden/amp_<name>.npz: the REFERENCE's training iteration under ``torch.autocast("cpu", bfloat16)``
    (engine.py:304,333-347) on the inputs of tests/golden/<name>.npz, and its fp32 gradients at the same sampled entries."""
    import os

    from tests.util import GOLDEN_DIR

    return np.load(os.path.join(GOLDEN_DIR, f"amp_{name}.npz"), allow_pickle=False)


def grad_sample_index(numel, n=512):
    # the entries oracle/make_golden.py keeps of a flattened gradient (restated: the generator itself imports the reference)
    if numel <= n:
        return np.arange(numel, dtype=np.int64)
    return np.unique(np.linspace(0, numel - 1, n).astype(np.int64))


@pytest.mark.parametrize("name,mode", [("train_b4", "bf16_train"), ("train_b4_t64", "bf16_train"),
                                       ("train_b4", "bf16_train_layerwise"), ("train_b4", "bf16_train_wgrad")])
def test_oracle_autocast_training_emulation_is_pinned_to_the_reference_autocast_iteration(name, mode):
    """``operand_rounding("bf16_train")`` (the fused kernels' rounding points), ``"bf16_train_layerwise"`` (those of the
    layer-wise autocast ops off the 64-point grid: fp32 rows between the GEMMs, fp32 feature transform) and
    ``"bf16_train_wgrad"`` (the same ops below 2048 rows - what ``train_b4``'s 896 rows take on the GPU: only the weight
    gradients are bf16-operand products) under autograd - what the HIP autocast training path is held to on the GPU - against the reference's own autocast iteration:
      * the fixture's fp32 samples are the fp32 oracle's, bit for bit (the fixture is the train_b4 iteration);
      * every loss term of the emulation within 3e-2 relative (+1e-3) of fp32 - the reference's bf16 losses are not (its
        loss arithmetic itself runs in bf16 on CPU autocast: ``loss_rot`` collapses to 0, ``loss_trans_z`` is 58 % off);
      * per tensor, on the sampled entries, the emulation is no further from the fp32 reference than the reference's own
        autocast run is (factor 1.2 + 5e-3 of the tensor's norm): rounding only GEMM operands and keeping GroupNorm, GELU,
        the pose update and the loss in fp32 is the milder of the two reduced-precision iterations."""
    from tests.util import load_train_golden

    g = load_train_golden(name)
    z = amp_fixture(name)
    torch.set_num_threads(4)
    l32, g32 = _oracle_train_iteration(g, None)
    lq, gq = _oracle_train_iteration(g, mode)
    assert len(gq) == 68 and set(gq) == set(g32)
    if mode != "bf16_train":   # it IS a different emulation: the two sets of rounding points do not coincide
        lf, gf = _oracle_train_iteration(g, "bf16_train")
        assert any(float((gq[k] - gf[k]).norm()) > 1e-4 * float(gf[k].norm()) for k in gq)
    for k in l32:
        np.testing.assert_allclose(l32[k], float(z[f"fp32__loss__{k}"][0]), rtol=2e-5, atol=1e-7, err_msg=k)
        assert abs(lq[k] - l32[k]) <= 3e-2 * abs(l32[k]) + 1e-3, (k, lq[k], l32[k])
    if name == "train_b4":   # what the docstring says
        assert float(z["loss__loss_rot"][0]) == 0.0 and float(z["fp32__loss__loss_rot"][0]) > 1e-3
    worse = []
    for k in g32:
        idx = torch.from_numpy(grad_sample_index(g32[k].numel()))
        r32, ramp = torch.from_numpy(z[f"fp32__gradsample__{k}"]), torch.from_numpy(z[f"gradsample__{k}"])
        np.testing.assert_allclose(g32[k].reshape(-1)[idx].numpy(), r32.numpy(), rtol=0, atol=2e-4 * float(r32.norm()) + 1e-9,
                                   err_msg=k)
        n = float(r32.norm()) + 1e-30
        d_emu, d_ref = float((gq[k].reshape(-1)[idx] - r32).norm()) / n, float((ramp - r32).norm()) / n
        if d_emu > 1.2 * d_ref + 5e-3:
            worse.append((k, d_emu, d_ref))
    assert not worse, worse


def test_teacher_forcing_with_the_runs_own_decisions_changes_nothing():
    """``oracle.teacher_forcing``: ReLU masks and max-pool winners taken from a recorded run.  Forced with the decisions of the
    very same run, the restatement returns the same outputs and the same gradients (the hooks sit where the decisions are
    taken and nowhere else); forced with the decisions of ANOTHER arithmetic (the bf16 emulation's), an fp32 run follows that
    activation pattern - its gradients move."""
    from tests.util import load_train_golden

    g = load_train_golden("train_b4_t64")
    cfg, b = g["cfg"], g["batch"]
    Gp = torch.randn(g["B"], 3, 4, generator=torch.Generator().manual_seed(1))

    def run(mode, force=None, record=None):
        sd = {k: v.clone().requires_grad_(True) for k, v in recipe_sd(cfg, g["salt"]).items()}
        ctx = O.teacher_forcing(record, record=True) if record is not None else O.teacher_forcing(force)
        with O.operand_rounding(mode), ctx:
            x, tfd = O.pose_apply(b["pcl"], b["obj_kps"], b["obj_pose_est"], b["obj_scale_est"])
            pose, scale = O.model_forward(x, tfd, b["obj_pose_est"], b["obj_scale_est"], sd, cfg, K_zoom=b["K"],
                                          mean_scales=b["obj_mean_scales"])
        ((pose * Gp).sum() + scale.sum()).backward()
        return pose.detach(), {k: p.grad for k, p in sd.items() if p.grad is not None}

    rec = {}
    p0, g0 = run(None, record=rec)
    assert {"x.pcl_net.stn.conv1", "k.pcl_net.fstn.pool", "x.pcl_net.pool", "x.flat", "k.pcl_net.conv3",
            "x.pcl_net.stn.poolrelu"} <= set(rec)
    p1, g1 = run(None, force=rec)
    assert torch.equal(p0, p1) and all(torch.equal(g0[k], g1[k]) for k in g0)
    recq = {}
    run("bf16_train", record=recq)
    assert any(not torch.equal(rec[k], recq[k]) for k in rec)          # the two arithmetics do decide differently somewhere
    p2, g2 = run(None, force=recq)
    assert any(not torch.equal(g0[k], g2[k]) for k in g0)
