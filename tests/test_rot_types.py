"""a10: the four rotation parametrisations of ``get_rot_mat`` (reference models/model_utils.py:28-40).

``tests/golden/rot_mats.npz`` holds outputs of the REFERENCE's own ``get_rot_mat`` / ``pose_scale_from_delta_init`` and
its autograd gradients (``oracle/make_golden.py rot_mats``).  CPU: the oracle restatement against them.  GPU: the HIP
conversions (``catre_rot_to_mat`` / ``_bwd``, and ``catre_pose_update`` / ``catre_op_pose_update_bwd`` with
``opts.rot_type``) against them, through the C ABI.
"""
import os

import numpy as np
import pytest
import torch

from oracle import catre_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rot_mats.npz")
TYPES = ("rot6d", "quat", "log_quat", "lie_vec")
DEV = "cuda:0"


def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("name", TYPES)
def test_oracle_rot_mat_matches_reference(name):
    z = gold()
    r = torch.from_numpy(z[f"{name}_in"]).requires_grad_(True)
    R = O.get_rot_mat(r, f"allo_{name}")
    np.testing.assert_allclose(R.detach().numpy(), z[f"{name}_R"], atol=1e-7, rtol=0)
    (R * torch.from_numpy(z["upstream"])).sum().backward()
    ref = z[f"{name}_grad"]
    ok = ~np.isnan(ref)
    assert np.array_equal(np.isnan(r.grad.numpy()), ~ok)  # same graph, same 0 * inf rows (lie_vec at exactly 0)
    np.testing.assert_allclose(r.grad.numpy()[ok], ref[ok], atol=1e-5, rtol=1e-5)
    # rotations: orthonormal, det +1 (the first-order lie_vec branch only to first order)
    Rn = z[f"{name}_R"].astype(np.float64)
    assert np.abs(Rn @ Rn.transpose(0, 2, 1) - np.eye(3)).max() < (4e-6 if name == "lie_vec" else 2e-6)  # theta^2 <= 1e-6 rows
    for allo in (False, True):
        Rt, tt, st = O.pose_scale_from_delta_init(
            torch.from_numpy(z[f"{name}_R"]), torch.from_numpy(z[f"{name}_dt"]), torch.from_numpy(z[f"{name}_ds"]),
            torch.from_numpy(z["R0"]), torch.from_numpy(z[f"{name}_t0"]), torch.from_numpy(z[f"{name}_s0"]),
            Ks=torch.from_numpy(z["K"]), K_aware=True, delta_T_space="image", is_allo=allo, scale_type="iter_add")
        tag = f"{name}_{'allo' if allo else 'ego'}"
        np.testing.assert_allclose(Rt.numpy(), z[f"{tag}_R"], atol=1e-6)
        np.testing.assert_allclose(tt.numpy(), z[f"{tag}_t"], atol=1e-6)
        np.testing.assert_allclose(st.numpy(), z[f"{tag}_s"], atol=1e-7)


def test_unknown_rot_type_raises_like_the_reference():
    from catre_amd.model_utils import get_rot_dim, get_rot_mat

    with pytest.raises(ValueError, match="Wrong pred_rot type"):
        get_rot_mat(torch.zeros(1, 6), "ego_euler")  # model_utils.py:39
    with pytest.raises(ValueError, match="Unknown rot_type"):
        get_rot_dim("ego_euler")  # model_utils.py:24
    assert [get_rot_dim(f"ego_{t}") for t in TYPES] == [6, 4, 3, 3]


# ------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", TYPES)
def test_hip_get_rot_mat_forward_and_backward_match_reference(name):
    from catre_amd.model_utils import get_rot_mat

    z = gold()
    r = torch.from_numpy(z[f"{name}_in"]).to(DEV).requires_grad_(True)
    R = get_rot_mat(r, f"ego_{name}")
    assert R.shape == (r.shape[0], 3, 3)
    np.testing.assert_allclose(R.detach().cpu().numpy(), z[f"{name}_R"], atol=2e-6, rtol=0)
    (R * torch.from_numpy(z["upstream"]).to(DEV)).sum().backward()
    got, ref = r.grad.cpu().numpy(), z[f"{name}_grad"]
    assert np.isfinite(got).all()
    ok = ~np.isnan(ref).any(1)
    scale = np.abs(ref[ok]).max(1, keepdims=True) + 1e-3
    assert (np.abs(got[ok] - ref[ok]) / scale).max() < 2e-4, (np.abs(got[ok] - ref[ok]) / scale).max()
    if name == "lie_vec":
        # where the reference's autograd gives 0 * inf = NaN (v == 0) the first-order branch's gradient is returned
        G = z["upstream"]
        bad = ~ok
        want = np.stack([G[bad, 2, 1] - G[bad, 1, 2], G[bad, 0, 2] - G[bad, 2, 0], G[bad, 1, 0] - G[bad, 0, 1]], 1)
        np.testing.assert_allclose(got[bad], want, atol=1e-6)
    # fp64 autograd of the oracle on the same inputs (tighter than the fp32 reference gradient)
    r64 = torch.from_numpy(z[f"{name}_in"]).double().requires_grad_(True)
    (O.get_rot_mat(r64, f"ego_{name}") * torch.from_numpy(z["upstream"]).double()).sum().backward()
    g64 = r64.grad.numpy()
    ok64 = np.isfinite(g64).all(1)
    scale = np.abs(g64[ok64]).max(1, keepdims=True) + 1e-3
    # tiny angles: fp32 cancellation in 1 - cos(theta) limits the HIP (and the reference's fp32) gradient
    big = ok64 & (np.linalg.norm(z[f"{name}_in"], axis=1) > 1e-2)
    assert (np.abs(got[big] - g64[big]) / (np.abs(g64[big]).max(1, keepdims=True) + 1e-3)).max() < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", TYPES)
@pytest.mark.parametrize("allo", [False, True])
def test_hip_pose_update_with_every_rot_type(name, allo):
    """``catre_pose_update`` fed with the raw residual (``opts.rot_type``) == the reference's get_rot_mat +
    pose_scale_from_delta_init; its backward vs fp64 autograd of the oracle."""
    from catre_amd import hip
    from catre_amd import train_ops as T

    z = gold()
    o = hip.CatreOpts()
    o.k_aware, o.refine_scale, o.delta_t_weight, o.allo_eps = 1, 1, 1.0, 1e-4
    o.is_allo = int(allo)
    o.rot_type = hip.rot_type_id(f"ego_{name}")
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    init_pose = torch.cat([dv(z["R0"]), dv(z[f"{name}_t0"]).reshape(-1, 3, 1)], -1)
    r, dt, ds = dv(z[f"{name}_in"]), dv(z[f"{name}_dt"]), dv(z[f"{name}_ds"])
    keep = np.linalg.norm(z[f"{name}_in"], axis=1) > 0  # v == 0: NaN gradient in the reference graph
    r, dt, ds, init_pose = r[keep], dt[keep], ds[keep], init_pose[keep]
    s0, K = dv(z[f"{name}_s0"])[keep], dv(z["K"])[keep]
    r.requires_grad_(True), dt.requires_grad_(True), ds.requires_grad_(True)
    pose, scale = T.pose_update_autograd(r, dt, ds, init_pose, s0, None, K, o)
    tag = f"{name}_{'allo' if allo else 'ego'}"
    np.testing.assert_allclose(pose[:, :, :3].detach().cpu().numpy(), z[f"{tag}_R"][keep], atol=3e-6)
    np.testing.assert_allclose(pose[:, :, 3].detach().cpu().numpy(), z[f"{tag}_t"][keep], atol=2e-6)
    np.testing.assert_allclose(scale.detach().cpu().numpy(), z[f"{tag}_s"][keep], atol=1e-7)
    gen = torch.Generator().manual_seed(3)
    Gp, Gs = torch.randn(pose.shape, generator=gen), torch.randn(scale.shape, generator=gen)
    ((pose * Gp.to(DEV)).sum() + (scale * Gs.to(DEV)).sum()).backward()
    # fp64 oracle autograd
    c64 = lambda a: torch.from_numpy(a[keep]).double()
    r64, dt64, ds64 = (c64(z[f"{name}_in"]).requires_grad_(True), c64(z[f"{name}_dt"]).requires_grad_(True),
                       c64(z[f"{name}_ds"]).requires_grad_(True))
    Rt, tt, st = O.pose_scale_from_delta_init(
        O.get_rot_mat(r64, f"ego_{name}"), dt64, ds64, c64(z["R0"]), c64(z[f"{name}_t0"]), c64(z[f"{name}_s0"]),
        Ks=c64(z["K"]), K_aware=True, delta_T_space="image", is_allo=allo, scale_type="iter_add")
    p64 = torch.cat([Rt, tt.reshape(-1, 3, 1)], -1)
    ((p64 * Gp.double()).sum() + (st * Gs.double()).sum()).backward()
    big = np.linalg.norm(z[f"{name}_in"][keep], axis=1) > 1e-2
    for got, want, what in ((r.grad, r64.grad, "d rot"), (dt.grad, dt64.grad, "d trans"), (ds.grad, ds64.grad, "d scale")):
        got, want = got.cpu().numpy()[big], want.numpy()[big]
        scale_ = np.abs(want).max(1, keepdims=True) + 1e-3
        assert (np.abs(got - want) / scale_).max() < 1e-4, (what, (np.abs(got - want) / scale_).max())


@pytest.mark.gpu
def test_quat_model_trains_with_gradients_matching_the_oracle():
    """ROT_TYPE=ego_quat (two rot heads of width rot_dim=2): the training path (padded neck, pose-update backward through
    quat2mat) vs fp64 autograd of the oracle on every parameter."""
    from catre_amd.CATRE_disR_shared import build_model_optimizer
    from tests.util import load_golden, recipe_sd

    g = load_golden("refine_b2_quat")
    cfg = g["cfg"].__deepcopy__({})
    cfg.MODEL.DEVICE = DEV
    model, _ = build_model_optimizer(cfg, is_test=False)
    sd = recipe_sd(cfg, g["salt"])
    assert sd["rot_head.rot_head_x.neck.0.weight"].shape == (2, 256, 1)
    model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=True)
    model.train()
    b = {k: v.to(DEV) for k, v in g["batch"].items()}
    from catre_amd.batching import batch_updater_test

    batch_updater_test(model.cfg, b, poses_est=None, scales_est=None)
    out = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                mean_scales=b["obj_mean_scales"], do_loss=False, cur_iter=1)
    assert np.abs(out["pose_1"].detach().cpu().numpy() - g["ref"]["pose_1"]).max() < 2e-5
    gen = torch.Generator().manual_seed(5)
    Gp, Gs = torch.randn(out["pose_1"].shape, generator=gen), torch.randn(out["scale_1"].shape, generator=gen)
    ((out["pose_1"] * Gp.to(DEV)).sum() + (out["scale_1"] * Gs.to(DEV)).sum()).backward()
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    bb = g["batch"]
    x, k = O.pose_apply(bb["pcl"].double(), bb["obj_kps"].double(), bb["obj_pose_est"].double(), bb["obj_scale_est"].double())
    pose, scale = O.model_forward(x, k, bb["obj_pose_est"].double(), bb["obj_scale_est"].double(), sd64, g["cfg"],
                                  K_zoom=bb["K"].double(), mean_scales=bb["obj_mean_scales"].double())
    ((pose * Gp.double()).sum() + (scale * Gs.double()).sum()).backward()
    for name, p in model.named_parameters():
        want = sd64[name].grad
        if ".norm." in name:
            assert p.grad is None and want is None
            continue
        got = p.grad.cpu().double()
        err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)
        assert err < 2e-4, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("rot_type,rot_dim", [("ego_log_quat", 3), ("ego_lie_vec", 3), ("ego_quat", 3), ("ego_rot6d", 2)])
def test_width_mismatch_is_rejected_at_construction(rot_type, rot_dim):
    """The reference's two-head ConvOutPerRotHead emits 2 x rot_dim values: a 3-wide residual cannot come out of it (the
    reference raises inside get_rot_mat at the first forward, pose_utils.py:357 / lie_algebra.py:23)."""
    from catre_amd.CATRE_disR_shared import build_model_optimizer
    from catre_amd.config import default_cfg

    cfg = default_cfg(num_pcl=64, num_kps=64, n_iter=1, device=DEV)
    cfg.MODEL.CATRE.ROT_HEAD.ROT_TYPE = rot_type
    cfg.MODEL.CATRE.ROT_HEAD.INIT_CFG.rot_dim = rot_dim
    with pytest.raises(ValueError, match="rot head emits"):
        build_model_optimizer(cfg, is_test=True)
