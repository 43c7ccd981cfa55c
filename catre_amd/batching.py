"""Loop glue either side of the model.

Test side: ``batch_updater_test`` (reference ``core/catre/engine/batch_test.py:63-99``).
Train side: ``batch_updater`` (``core/catre/engine/batching.py:92-146``), the initial estimates
``get_init_pose_train`` / ``get_init_scale_train`` / ``get_normed_kps`` (``engine_utils.py:17-35,187-247``) and the
batch augmentations ``aug_3d_bbox`` / ``aug_RT`` (``engine_utils.py:107-172``) that ``batch_data`` applies
(``batching.py:84-88``).  Same names, arguments and random-number consumption as the reference; the per-object Python
loops of tiny ``torch.mm`` calls are single HIP launches (``catre_aug_points``, ``catre_init_noise``,
``catre_pose_apply``).
"""
import ctypes
import math
import random

import numpy as np
import torch

from . import hip
from .pose_aug import aug_poses_normal, aug_scale_normal
from .runtime import pose_apply


def batch_updater_test(cfg, batch, poses_est=None, scales_est=None, device="cuda", dtype=torch.float32):
    """In-place update of ``batch`` for the next refine iteration: feeds back the estimates and
    writes ``batch["tfd_kps"] [B,3,M]`` and ``batch["x"] [B,3,N]`` (permuted views, like the reference)."""
    if poses_est is not None:
        batch["obj_pose_est"] = poses_est
    if scales_est is not None and cfg.MODEL.REFINE_SCLAE:
        batch["obj_scale_est"] = scales_est
    if "obj_kps" not in batch:
        raise KeyError("batch['obj_kps'] missing (the reference fills it with get_normed_kps, engine_utils.py:17-35)")
    x, tfd_kps = pose_apply(batch["pcl"], batch["obj_kps"], batch["obj_pose_est"], batch["obj_scale_est"],
                            zero_center=cfg.INPUT.ZERO_CENTER_INPUT)
    batch["tfd_kps"] = tfd_kps
    batch["x"] = x


# ------------------------------------------------------------------------------------------ augmentation
def _sym_flags(batch, device):
    sym = batch.get("sym_info")
    if sym is None:
        return None
    return torch.tensor([0 if s is None else 1 for s in sym], dtype=torch.int32, device=device)


def _aug_points(batch, ratios=None, delta_r=None, delta_t=None):
    lib = hip.load()
    pcl = hip.require_dev_f32(batch["pcl"].contiguous(), "pcl")
    B, N = pcl.shape[0], pcl.shape[1]
    dev = pcl.device
    pose = hip.require_dev_f32(batch["obj_pose"].contiguous(), "obj_pose", (B, 3, 4))
    scale = hip.require_dev_f32(batch["obj_scale"].contiguous(), "obj_scale", (B, 3))
    flags = _sym_flags(batch, dev) if ratios is not None else None
    f3, f9 = ctypes.c_float * 3, ctypes.c_float * 9
    c_ratios = f3(*[float(v) for v in ratios]) if ratios is not None else None
    c_dr = f9(*[float(v) for v in delta_r.reshape(-1)]) if delta_r is not None else None
    c_dt = f3(*[float(v) for v in delta_t.reshape(-1)]) if delta_t is not None else None
    pcl_out, pose_out, scale_out = torch.empty_like(pcl), torch.empty_like(pose), torch.empty_like(scale)
    hip.check(lib.catre_aug_points(hip.ptr(pcl), hip.ptr(pose), hip.ptr(scale), hip.ptr(flags), c_ratios, c_dr, c_dt,
                                   hip.ptr(pcl_out), hip.ptr(pose_out), hip.ptr(scale_out), B, N, hip.stream_ptr(dev)),
              "catre_aug_points")
    batch["pcl"], batch["obj_pose"], batch["obj_scale"] = pcl_out, pose_out, scale_out


def aug_3d_bbox(batch, shift_sx=(0.8, 1.2), shift_sy=(0.8, 1.2), shift_sz=(0.8, 1.2), device="cuda",
                dtype=torch.float32):
    """``engine_utils.py:107-139``: stretch every object along its own axes (x and z tied for y-symmetric ones)."""
    ex, ey, ez = torch.rand(3)
    ex = ex * (shift_sx[1] - shift_sx[0]) + shift_sx[0]
    ey = ey * (shift_sy[1] - shift_sy[0]) + shift_sy[0]
    ez = ez * (shift_sz[1] - shift_sz[0]) + shift_sz[0]
    _aug_points(batch, ratios=(ex, ey, ez))


def get_rotation_torch(x_, y_, z_):
    """``engine_utils.py:175-184`` (degrees in, R_z R_y R_x out)."""
    x, y, z = (float(v) / 180 * math.pi for v in (x_, y_, z_))
    R_x = torch.tensor([[1, 0, 0], [0, math.cos(x), -math.sin(x)], [0, math.sin(x), math.cos(x)]])
    R_y = torch.tensor([[math.cos(y), 0, math.sin(y)], [0, 1, 0], [-math.sin(y), 0, math.cos(y)]])
    R_z = torch.tensor([[math.cos(z), -math.sin(z), 0], [math.sin(z), math.cos(z), 0], [0, 0, 1]])
    return torch.mm(R_z, torch.mm(R_y, R_x))


def aug_RT(batch, shift_tx=0.005, shift_ty=0.005, shift_tz=0.025, shift_rot=15.0, device="cuda", dtype=torch.float32):
    """``engine_utils.py:142-172``: one random rigid motion applied to every cloud and pose of the batch."""
    rx, ry, rz = torch.rand(3) * shift_rot * 2 - shift_rot
    tx = torch.rand(1) * shift_tx * 2 - shift_tx
    ty = torch.rand(1) * shift_ty * 2 - shift_ty
    tz = torch.rand(1) * shift_tz * 2 - shift_tz
    delta_r = get_rotation_torch(rx, ry, rz).to(torch.float32)
    delta_t = torch.tensor((tx, ty, tz), dtype=torch.float32)
    _aug_points(batch, delta_r=delta_r, delta_t=delta_t)


def apply_train_augmentation(cfg, batch):
    """The tail of ``batch_data`` (``batching.py:84-88``)."""
    if torch.rand(1) < cfg.INPUT.BBOX3D_AUG_PROB:
        aug_3d_bbox(batch)
    if torch.rand(1) < cfg.INPUT.RT_AUG_PROB:
        aug_RT(batch)
    return batch


# ------------------------------------------------------------------------------------------ initial estimates
def _axangle_chain(chain):
    """``rot_from_axangle_chain`` (``core/utils/pose_utils.py:31-35``): product of axis-angle rotations, the angle
    given in units of pi."""
    R = np.eye(3)
    for ax_x, ax_y, ax_z, frac in chain:
        ang = math.pi * frac
        ax = np.array([ax_x, ax_y, ax_z], dtype=np.float64)
        ax = ax / np.linalg.norm(ax)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = R @ (np.eye(3) + math.sin(ang) * Kx + (1 - math.cos(ang)) * (Kx @ Kx))
    return R


def get_init_pose_train(cfg, batch, device="cuda", dtype=torch.float32):
    """``engine_utils.py:213-247``."""
    input_cfg = cfg.INPUT
    n_obj = batch["obj_pose"].shape[0]
    kw = dict(dtype=dtype, device=batch["obj_pose"].device)
    init_pose_type = random.choice(input_cfg.INIT_POSE_TYPE_TRAIN)
    if init_pose_type == "gt_noise":
        batch["obj_pose_est"] = aug_poses_normal(batch["obj_pose"], std_rot=input_cfg.NOISE_ROT_STD_TRAIN,
                                                 std_trans=input_cfg.NOISE_TRANS_STD_TRAIN,
                                                 max_rot=input_cfg.NOISE_ROT_MAX_TRAIN, min_z=input_cfg.INIT_TRANS_MIN_Z)
    elif init_pose_type == "random":
        # uniformly random rotations (unit quaternions from normals) and translations inside the configured box
        q = torch.randn(n_obj, 4)
        q = q / q.norm(dim=1, keepdim=True)
        w, x, y, z = q.unbind(1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n_obj, 3, 3)
        lo, hi = torch.tensor(input_cfg.RANDOM_TRANS_MIN), torch.tensor(input_cfg.RANDOM_TRANS_MAX)
        t = lo + (hi - lo) * torch.rand(n_obj, 3)
        batch["obj_pose_est"] = torch.cat([R, t.unsqueeze(-1)], -1).to(**kw)
    elif init_pose_type == "last_frame":
        assert "last_frame_poses" in batch
        batch["obj_pose_est"] = batch["last_frame_poses"][:, :3, :4]
    elif init_pose_type == "canonical":
        pose = np.hstack([_axangle_chain(input_cfg.CANONICAL_ROT), np.array(input_cfg.CANONICAL_TRANS).reshape(3, 1)])
        batch["obj_pose_est"] = torch.tensor(pose, **kw).repeat(n_obj, 1, 1)
    else:
        raise ValueError(f"Unknown init pose type for train: {init_pose_type}")


def get_init_scale_train(cfg, batch, device="cuda", dtype=torch.float32):
    """``engine_utils.py:187-210``."""
    input_cfg = cfg.INPUT
    n_obj = batch["obj_scale"].shape[0]
    kw = dict(dtype=dtype, device=batch["obj_scale"].device)
    init_type = random.choice(input_cfg.INIT_SCALE_TYPE_TRAIN)
    if init_type == "gt_noise":
        batch["obj_scale_est"] = aug_scale_normal(batch["obj_scale"], std_scale=input_cfg.NOISE_SCALE_STD_TRAIN,
                                                  min_s=input_cfg.INIT_SCALE_MIN)
    elif init_type == "random":
        lo, hi = torch.tensor(input_cfg.RANDOM_SCALE_MIN), torch.tensor(input_cfg.RANDOM_SCALE_MAX)
        batch["obj_scale_est"] = (lo + (hi - lo) * torch.rand(n_obj, 3)).to(**kw)
    elif init_type == "last_frame":
        batch["obj_scale_est"] = batch["last_frame_poses"][:, :3, 4]
    elif init_type == "canonical":
        batch["obj_scale_est"] = torch.tensor(input_cfg.CANONICAL_SIZE, **kw).reshape(1, 3).repeat(n_obj, 1)
    else:
        raise ValueError(f"Unknown init pose type for train: {init_type}")


def get_normed_bbox(bs):
    h = 0.5
    box = torch.tensor([[h, h, h], [-h, h, h], [-h, -h, h], [h, -h, h], [h, h, -h], [-h, h, -h], [-h, -h, -h], [h, -h, -h]])
    return box.unsqueeze(0).repeat(bs, 1, 1)


def get_normed_axis(bs, num_kps=4, with_neg=False):
    n = (num_kps - 1) // 3
    start, length = (-0.5, 1.0) if with_neg else (0.0, 0.5)
    steps = torch.tensor([start + length * i / n for i in range(1, n + 1)])
    pts = torch.zeros(3 * n + 1, 3)
    for a in range(3):
        pts[a * n:(a + 1) * n, a] = steps
    return pts.unsqueeze(0).repeat(bs, 1, 1)


def get_normed_kps(cfg, batch, **to_float_args):
    """``engine_utils.py:17-35``: the prior / keypoint set ``batch["obj_kps"] [B,M,3]``."""
    kps_type = cfg.INPUT.KPS_TYPE.lower()
    dev = batch["obj_scale_est"].device
    if kps_type == "bbox":
        batch["obj_kps"] = get_normed_bbox(batch["obj_scale_est"].shape[0]).to(dev)
    elif kps_type == "mean_shape":
        batch["obj_kps"] = batch["obj_mean_points"].clone()
    elif kps_type == "fps":
        batch["obj_kps"] = batch["obj_fps_points"] / batch["obj_scale_est"].unsqueeze(1)
    elif kps_type == "axis":
        batch["obj_kps"] = get_normed_axis(batch["obj_scale_est"].shape[0], cfg.INPUT.NUM_KPS,
                                           cfg.INPUT.WITH_NEG_AXIS).to(dev)
    else:
        raise NotImplementedError(f"Unknown keypoints type {kps_type}")


def batch_updater(cfg, batch, cur_iter=1, poses_est=None, scales_est=None, device="cuda", dtype=torch.float32,
                  phase="train"):
    """``core/catre/engine/batching.py:92-146``."""
    if phase == "test":
        return batch_updater_test(cfg, batch, poses_est=poses_est, scales_est=scales_est, device=device, dtype=dtype)
    if poses_est is None:
        get_init_pose_train(cfg, batch)
    else:
        batch["obj_pose_est"] = poses_est
    if scales_est is None:
        if cfg.MODEL.REFINE_SCLAE:
            get_init_scale_train(cfg, batch)
        else:
            batch["obj_scale_est"] = batch["obj_scale"].detach().clone()
    else:
        batch["obj_scale_est"] = scales_est
    if "obj_kps" not in batch:
        get_normed_kps(cfg, batch)
    x, tfd_kps = pose_apply(batch["pcl"], batch["obj_kps"], batch["obj_pose_est"], batch["obj_scale_est"],
                            zero_center=cfg.INPUT.ZERO_CENTER_INPUT)
    batch["tfd_kps"] = tfd_kps
    batch["x"] = x
