"""Initial-pose / initial-scale noise on the device - counterparts of the reference's ``core/utils/pose_aug.py``.

``aug_poses_normal`` (``:59-101``) and ``aug_scale_normal`` (``:10-35``) keep their signatures and draw their random
numbers with exactly the same ``numpy`` / ``torch`` calls in the same order (so a seeded run consumes the generators
identically); the arithmetic - clamp, Euler angles -> rotation (``pose_utils.py:266-296``), compose, z / range clamps -
is one ``catre_init_noise`` launch instead of ~25 small torch kernels.
"""
from collections.abc import Sequence

import numpy as np
import torch

from . import hip


def _is_nested(seq):
    return isinstance(seq, (tuple, list, Sequence)) and len(seq) > 0 and isinstance(seq[0], (tuple, list, Sequence))


def init_noise(poses=None, euler_deg=None, trans_noise=None, max_rot=None, min_z=0.1, scales=None, scale_noise=None,
               min_s=0.04, max_s=0.45):
    """The deterministic part: noise in, perturbed ``poses [B,3,4]`` and / or ``scales [B,3]`` out."""
    lib = hip.load()
    ref = poses if poses is not None else scales
    B, dev = ref.shape[0], ref.device
    pose_out = scale_out = None
    if poses is not None:
        poses = hip.require_dev_f32(poses.contiguous(), "poses", (B, 3, 4))
        euler_deg = hip.require_dev_f32(euler_deg.contiguous(), "euler_deg", (B, 3))
        trans_noise = hip.require_dev_f32(trans_noise.contiguous(), "trans_noise", (B, 3))
        pose_out = torch.empty_like(poses)
    if scales is not None:
        scales = hip.require_dev_f32(scales.contiguous(), "scales", (B, 3))
        scale_noise = hip.require_dev_f32(scale_noise.contiguous(), "scale_noise", (B, 3))
        scale_out = torch.empty_like(scales)
    hip.check(lib.catre_init_noise(hip.ptr(poses), hip.ptr(euler_deg), hip.ptr(trans_noise),
                                   -1.0 if max_rot is None else float(max_rot), float(min_z), hip.ptr(pose_out),
                                   hip.ptr(scales), hip.ptr(scale_noise), float(min_s), float(max_s), hip.ptr(scale_out),
                                   B, hip.stream_ptr(dev)), "catre_init_noise")
    return pose_out, scale_out


def aug_scale_normal(scales, std_scale=[0.11, 0.04, 0.9], min_s=0.04, max_s=0.45):
    device = scales.device
    if _is_nested(std_scale):
        sel_std_scale = std_scale[np.random.choice(len(std_scale))]  # randomly choose one setting (:20-23)
    else:
        sel_std_scale = std_scale
    scale_noises = torch.normal(mean=torch.zeros_like(scales),
                                std=torch.tensor(sel_std_scale, device=device).view(1, 3))
    return init_noise(scales=scales, scale_noise=scale_noises, min_s=min_s, max_s=max_s)[1]


def aug_poses_normal(poses, std_rot=15, std_trans=[0.01, 0.01, 0.05], max_rot=45, min_z=0.1):
    assert poses.ndim == 3, poses.shape
    bs, device = poses.shape[0], poses.device
    if isinstance(std_rot, (tuple, list, Sequence)):
        std_rot = np.random.choice(std_rot)
    euler_noises_deg = torch.normal(mean=0, std=std_rot, size=(bs, 3)).to(device=device)  # CPU generator, like :79
    if _is_nested(std_trans):
        sel_std_trans = std_trans[np.random.choice(len(std_trans))]
    else:
        sel_std_trans = std_trans
    trans_noises = torch.normal(mean=torch.zeros_like(poses[:, :3, 3]),
                                std=torch.tensor(sel_std_trans, device=device).view(1, 3))
    return init_noise(poses=poses, euler_deg=euler_noises_deg, trans_noise=trans_noises, max_rot=max_rot, min_z=min_z)[0]
