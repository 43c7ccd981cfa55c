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
