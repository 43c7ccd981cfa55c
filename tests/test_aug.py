"""Row f2: train-time batch glue (augmentation + initial-estimate noise).  The golden holds outputs of the REFERENCE
functions (engine_utils.aug_3d_bbox / aug_RT, pose_aug.aug_poses_normal / aug_scale_normal) together with the random
draws they consumed.  Tolerance: 2e-6 abs (fp32 re-association of 3x3 products; values are O(1))."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import aug_oracle as AO
from tests.util import GOLDEN_DIR

TOL = 2e-6
DEV = "cuda:0"


def _g():
    z = np.load(os.path.join(GOLDEN_DIR, "aug_train.npz"))
    return {k: z[k] for k in z.files}


def _t(a, dev="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_aug_oracle_matches_reference_functions():
    g = _g()
    p1, s1 = AO.aug_3d_bbox(_t(g["in_pcl"]), _t(g["in_pose"]), _t(g["in_scale"]), _t(g["in_sym"]), g["bbox_ratios"])
    assert np.abs(p1.numpy() - g["bbox_pcl"]).max() < TOL and np.abs(s1.numpy() - g["bbox_scale"]).max() < TOL
    p2, q2 = AO.aug_rt(p1, _t(g["in_pose"]), _t(g["rt_delta_r"]), _t(g["rt_delta_t"]))
    assert np.abs(p2.numpy() - g["rt_pcl"]).max() < TOL and np.abs(q2.numpy() - g["rt_pose"]).max() < TOL
    pn = AO.poses_from_noise(_t(g["noise_pose_in"]), _t(g["noise_euler_deg"]), _t(g["noise_trans"]), 1.0, 0.1)
    assert np.abs(pn.numpy() - g["noise_pose_out"]).max() < TOL
    assert pn[0, 2, 3] == pytest.approx(0.1) and np.abs(g["noise_euler_deg"]).max() > 1.0  # both clamps exercised
    sn = AO.scales_from_noise(_t(g["noise_scale_in"]), _t(g["noise_scale"]), 0.04, 0.45)
    assert np.abs(sn.numpy() - g["noise_scale_out"]).max() < TOL


def test_canonical_rotation_chain():
    from scipy.spatial.transform import Rotation

    from catre_amd.batching import _axangle_chain

    chain = [(1, 0, 0, 0.5), (0, 0, 1, -0.7)]  # configs/_base_/catre_base.py:84
    want = Rotation.from_rotvec([0.5 * np.pi, 0, 0]).as_matrix() @ Rotation.from_rotvec([0, 0, -0.7 * np.pi]).as_matrix()
    assert np.abs(_axangle_chain(chain) - want).max() < 1e-12


def _batch(g):
    return {"pcl": _t(g["in_pcl"], DEV), "obj_pose": _t(g["in_pose"], DEV), "obj_scale": _t(g["in_scale"], DEV),
            "sym_info": [np.eye(3)[None] if f else None for f in g["in_sym"]]}


@pytest.mark.gpu
def test_hip_augmentation_matches_reference_with_same_seed():
    """aug_3d_bbox / aug_RT draw their parameters with the reference's own torch.rand calls: same seed, same result."""
    from catre_amd import batching

    g = _g()
    batch = _batch(g)
    pcl_in = batch["pcl"].clone()
    torch.manual_seed(77)
    batching.aug_3d_bbox(batch)
    assert np.abs(batch["pcl"].cpu().numpy() - g["bbox_pcl"]).max() < TOL
    assert np.abs(batch["obj_scale"].cpu().numpy() - g["bbox_scale"]).max() < TOL
    assert np.abs(batch["obj_pose"].cpu().numpy() - g["in_pose"]).max() == 0
    torch.manual_seed(77)
    batching.aug_RT(batch)
    assert np.abs(batch["pcl"].cpu().numpy() - g["rt_pcl"]).max() < TOL
    assert np.abs(batch["obj_pose"].cpu().numpy() - g["rt_pose"]).max() < TOL
    assert torch.equal(pcl_in, _t(g["in_pcl"], DEV)), "the caller's tensors are not modified in place"


@pytest.mark.gpu
def test_hip_fused_bbox_and_rt_single_launch():
    """Both augmentations in one catre_aug_points launch == the reference's two passes."""
    import ctypes

    from catre_amd import hip

    g = _g()
    pcl, pose, scale = _t(g["in_pcl"], DEV), _t(g["in_pose"], DEV), _t(g["in_scale"], DEV)
    sym = _t(g["in_sym"], DEV)
    B, N = pcl.shape[:2]
    po, qo, so = torch.empty_like(pcl), torch.empty_like(pose), torch.empty_like(scale)
    f3, f9 = ctypes.c_float * 3, ctypes.c_float * 9
    hip.check(hip.load().catre_aug_points(hip.ptr(pcl), hip.ptr(pose), hip.ptr(scale), hip.ptr(sym), f3(*g["bbox_ratios"]),
                                          f9(*g["rt_delta_r"].reshape(-1)), f3(*g["rt_delta_t"]), hip.ptr(po), hip.ptr(qo),
                                          hip.ptr(so), B, N, hip.stream_ptr(pcl.device)), "catre_aug_points")
    assert np.abs(po.cpu().numpy() - g["rt_pcl"]).max() < TOL
    assert np.abs(qo.cpu().numpy() - g["rt_pose"]).max() < TOL
    assert np.abs(so.cpu().numpy() - g["bbox_scale"]).max() < TOL
    # bad arguments are refused, not executed
    assert hip.load().catre_aug_points(hip.ptr(pcl), hip.ptr(pose), hip.ptr(scale), None, None, f9(*([0.0] * 9)), None,
                                       hip.ptr(po), hip.ptr(qo), hip.ptr(so), B, N, None) != 0


@pytest.mark.gpu
def test_hip_init_noise_matches_reference():
    from catre_amd import pose_aug

    g = _g()
    pn, sn = pose_aug.init_noise(poses=_t(g["noise_pose_in"], DEV), euler_deg=_t(g["noise_euler_deg"], DEV),
                                 trans_noise=_t(g["noise_trans"], DEV), max_rot=1.0, min_z=0.1,
                                 scales=_t(g["noise_scale_in"], DEV), scale_noise=_t(g["noise_scale"], DEV), min_s=0.04)
    assert np.abs(pn.cpu().numpy() - g["noise_pose_out"]).max() < TOL
    assert np.abs(sn.cpu().numpy() - g["noise_scale_out"]).max() < TOL
    # the public functions consume numpy / CPU-torch generators like the reference: the rotation part (CPU normal)
    # reproduces the reference exactly under the same seed; the translation noise comes from the device generator
    torch.manual_seed(77); np.random.seed(77); random.seed(77)
    out = pose_aug.aug_poses_normal(_t(g["noise_pose_in"], DEV), std_rot=(10, 5, 2.5, 1.25),
                                    std_trans=[(0.02, 0.02, 0.02), (0.01, 0.01, 0.01), (0.005, 0.005, 0.005)], max_rot=1.0)
    assert np.abs(out[:, :, :3].cpu().numpy() - g["noise_pose_out"][:, :, :3]).max() < TOL
    assert out[0, 2, 3].item() == pytest.approx(0.1)
    torch.manual_seed(5); np.random.seed(5)
    a = pose_aug.aug_scale_normal(_t(g["noise_scale_in"], DEV), std_scale=[(0.01, 0.01, 0.01), (0.005, 0.005, 0.005)])
    torch.manual_seed(5); np.random.seed(5)
    b = pose_aug.aug_scale_normal(_t(g["noise_scale_in"], DEV), std_scale=[(0.01, 0.01, 0.01), (0.005, 0.005, 0.005)])
    assert torch.equal(a, b) and a.min().item() >= 0.04 and not torch.equal(a, _t(g["noise_scale_in"], DEV))


@pytest.mark.gpu
def test_train_batch_updater_drives_a_training_iteration():
    """batch_data tail + batch_updater (train) + model(do_loss=True): the reference's loop body engine.py:289-330."""
    from catre_amd import batching, synth
    from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
    from catre_amd.config import default_cfg

    cfg = default_cfg(num_pcl=128, num_kps=96, device=DEV)
    cfg.INPUT.BBOX3D_AUG_PROB = cfg.INPUT.RT_AUG_PROB = 1.0
    model, opt = build_model_optimizer(cfg, is_test=False)
    model.load_state_dict({k: v.to(DEV) for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
    inp = {k: v.to(DEV) for k, v in synth.make_inputs(4, 128, 96, seed=3).items()}
    batch = {"pcl": inp["pcl"], "obj_pose": torch.cat([inp["gt_rot"], inp["gt_trans"].unsqueeze(-1)], -1),
             "obj_scale": inp["gt_scale"], "obj_mean_points": inp["obj_kps"], "obj_mean_scales": inp["obj_mean_scales"],
             "K": inp["K"], "sym_info": [None, np.eye(3)[None], None, None]}
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    batching.apply_train_augmentation(cfg, batch)
    poses_est = scales_est = None
    for it in range(1, 3):
        batching.batch_updater(cfg, batch, cur_iter=it, poses_est=poses_est, scales_est=scales_est)
        assert batch["x"].shape == (4, 3, 128) and batch["tfd_kps"].shape == (4, 3, 96)
        if it == 1:  # gt + noise: close to, but not equal to, the ground truth
            dt = (batch["obj_pose_est"][:, :, 3] - batch["obj_pose"][:, :, 3]).abs().max().item()
            assert 0 < dt < 0.1
            assert torch.allclose(batch["x"], batch["pcl"].permute(0, 2, 1) - batch["obj_pose_est"][:, :, 3:4], atol=1e-6)
        out, losses = model(batch["x"], batch["tfd_kps"], init_pose=batch["obj_pose_est"], init_scale=batch["obj_scale_est"],
                            K_zoom=batch["K"], gt_ego_rot=batch["obj_pose"][:, :3, :3], gt_trans=batch["obj_pose"][:, :3, 3],
                            gt_scale=batch["obj_scale"], obj_kps=batch["obj_kps"], mean_scales=batch["obj_mean_scales"],
                            sym_info=batch["sym_info"], do_loss=True, cur_iter=it)
        loss = sum(losses.values())
        assert torch.isfinite(loss)
        loss.backward()
        opt.step(); opt.zero_grad(set_to_none=True)
        poses_est, scales_est = out[f"pose_{it}"].detach(), out[f"scale_{it}"].detach()
    for typ in ("canonical", "random"):
        cfg.INPUT.INIT_POSE_TYPE_TRAIN = [typ]
        cfg.INPUT.INIT_SCALE_TYPE_TRAIN = [typ]
        batching.batch_updater(cfg, batch)
        R = batch["obj_pose_est"][:, :, :3]
        assert (R @ R.transpose(1, 2) - torch.eye(3, device=DEV)).abs().max() < 1e-5
        assert batch["obj_scale_est"].shape == (4, 3) and (batch["obj_scale_est"] > 0).all()
    cfg.INPUT.INIT_POSE_TYPE_TRAIN = ["nope"]
    with pytest.raises(ValueError):
        batching.batch_updater(cfg, batch)
