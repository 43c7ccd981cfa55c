"""Pose / scale update - same name, arguments and error behaviour as reference
``core/catre/models/pose_scale_from_delta_init.py:8-95``; arithmetic in ``catre_pose_update`` (HIP)."""
import torch

from . import hip
from .runtime import pose_update


def pose_scale_from_delta_init(
    rot_deltas,
    trans_deltas,
    scale_deltas,
    rot_inits,
    trans_inits,
    scale_inits,
    Ks=None,
    K_aware=False,
    delta_T_space="3D",
    delta_T_weight=1.0,
    delta_z_style="cosypose",
    eps=1e-4,
    is_allo=False,
    scale_type="add_iter",
):
    """rot_deltas [b,3,3] (already a rotation matrix), trans_deltas / scale_deltas [b,3] ->
    (rot_tgts [b,3,3], trans_tgts [b,3], scale_tgts [b,3])."""
    bs = rot_deltas.shape[0]
    assert rot_deltas.shape == (bs, 3, 3)
    assert rot_inits.shape == (bs, 3, 3)
    assert trans_deltas.shape == (bs, 3)
    assert trans_inits.shape == (bs, 3)
    if delta_T_space not in ("image", "3D"):
        raise ValueError("Unknown delta_T_space: {}".format(delta_T_space))
    if delta_T_space == "image" and K_aware:
        assert Ks is not None and Ks.shape == (bs, 3, 3)
    o = hip.CatreOpts()
    o.delta_t_space_3d = int(delta_T_space == "3D")
    o.delta_z_deepim = int(delta_z_style != "cosypose")
    o.k_aware = int(bool(K_aware))
    o.scale_mul = int("add" not in scale_type)
    o.scale_base_mean = 0  # the caller already chose the base (scale_inits)
    o.is_allo = int(bool(is_allo))
    o.refine_scale = 1
    o.delta_t_weight = float(delta_T_weight)
    o.allo_eps = float(eps)
    o.rot_input_is_matrix = 1
    rot6d = rot_deltas.contiguous()
    init_pose = torch.cat([rot_inits, trans_inits.reshape(bs, 3, 1)], dim=-1)
    pose, scale = pose_update(rot6d, trans_deltas, scale_deltas, init_pose, scale_inits, None, Ks, o)
    return pose[:, :3, :3], pose[:, :3, 3], scale
