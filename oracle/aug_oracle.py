"""ORACLE (test infrastructure only) - batched restatement of the reference's train-time batch glue with the random
draws made explicit: ``aug_3d_bbox`` / ``aug_RT`` (``core/catre/engine/engine_utils.py:107-172``),
``euler2mat_torch`` (``core/utils/pose_utils.py:266-296``), ``aug_poses_normal`` / ``aug_scale_normal``
(``core/utils/pose_aug.py:10-101``).  Pinned to the reference functions by ``tests/golden/aug_train.npz``."""
import math

import torch


def aug_3d_bbox(pcl, pose, scale, sym_flags, ratios):
    """pcl [B,N,3], pose [B,3,4], scale [B,3], sym_flags [B] bool, ratios (ex,ey,ez) -> pcl', scale'."""
    R, t = pose[:, :, :3], pose[:, :, 3]
    ex, ey, ez = (torch.as_tensor(v, dtype=pcl.dtype) for v in ratios)
    exz = (ex + ez) / 2
    r = torch.where(sym_flags.reshape(-1, 1).bool(), torch.stack([exz, ey, exz]).reshape(1, 3),
                    torch.stack([ex, ey, ez]).reshape(1, 3))  # [B,3]
    q = torch.einsum("bji,bnj->bni", R, pcl - t.unsqueeze(1)) * r.unsqueeze(1)  # R^T (p - t), scaled
    return torch.einsum("bij,bnj->bni", R, q) + t.unsqueeze(1), scale * r


def aug_rt(pcl, pose, delta_r, delta_t):
    R, t = pose[:, :, :3], pose[:, :, 3]
    pcl2 = torch.einsum("ij,bnj->bni", delta_r, pcl + delta_t.reshape(1, 1, 3))
    R2 = torch.einsum("ij,bjk->bik", delta_r, R)
    t2 = torch.einsum("ij,bj->bi", delta_r, t + delta_t.reshape(1, 3))
    return pcl2, torch.cat([R2, t2.unsqueeze(-1)], -1)


def euler2mat(angle):
    x, y, z = angle[:, 0], angle[:, 1], angle[:, 2]
    zero, one = torch.zeros_like(x), torch.ones_like(x)
    zm = torch.stack([z.cos(), -z.sin(), zero, z.sin(), z.cos(), zero, zero, zero, one], 1).reshape(-1, 3, 3)
    ym = torch.stack([y.cos(), zero, y.sin(), zero, one, zero, -y.sin(), zero, y.cos()], 1).reshape(-1, 3, 3)
    xm = torch.stack([one, zero, zero, zero, x.cos(), -x.sin(), zero, x.sin(), x.cos()], 1).reshape(-1, 3, 3)
    return xm @ ym @ zm


def poses_from_noise(poses, euler_deg, trans_noise, max_rot=45, min_z=0.1):
    e = euler_deg if max_rot is None else euler_deg.clamp(-max_rot, max_rot)
    out = poses.clone()
    out[:, :3, :3] = euler2mat(e * math.pi / 180.0) @ poses[:, :3, :3]
    out[:, :3, 3] = poses[:, :3, 3] + trans_noise
    out[:, 2, 3] = out[:, 2, 3].clamp(min=max(min_z, 1e-4))
    return out


def scales_from_noise(scales, noise, min_s=0.04, max_s=0.45):
    return (scales + noise).clamp(min=max(min_s, 1e-4), max=max_s)
