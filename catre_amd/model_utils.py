"""Head builders - same names/returns as reference ``core/catre/models/model_utils.py:66-89,144-167``."""
import copy

import torch

from . import hip
from .net_factory import HEADS, PCLNETS  # noqa: F401


def get_rot_dim(rot_type):
    """reference model_utils.py:11-25."""
    if rot_type in ["allo_quat", "ego_quat"]:
        return 4
    if rot_type in ["allo_log_quat", "ego_log_quat", "allo_lie_vec", "ego_lie_vec"]:
        return 3
    if rot_type in ["allo_rot6d", "ego_rot6d"]:
        return 6
    raise ValueError(f"Unknown rot_type: {rot_type}")


class _RotToMat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rot, type_id):
        from .runtime import rot_to_mat

        rot = rot.contiguous()
        ctx.save_for_backward(rot)
        ctx.type_id = type_id
        return rot_to_mat(rot, type_id)

    @staticmethod
    def backward(ctx, grad_R):
        from .runtime import rot_to_mat_bwd

        (rot,) = ctx.saved_tensors
        return rot_to_mat_bwd(rot, ctx.type_id, grad_R), None


def get_rot_mat(rot, rot_type):
    """reference model_utils.py:28-40: [B,d] residual -> [B,3,3] rotation matrix for ``{ego,allo}_{quat,log_quat,lie_vec,
    rot6d}`` - ``quat2mat_torch`` (pose_utils.py:349-412), ``quat2mat_torch(qexp(.))`` (quaternion_lf.py:294-317),
    ``lie_vec_to_rot`` (lie_algebra.py:7-77), ``rot6d_to_mat_batch`` (rot_reps.py:34-55) - one HIP launch
    (``catre_rot_to_mat``), autograd-connected (``catre_rot_to_mat_bwd``)."""
    try:
        type_id = hip.rot_type_id(rot_type)
    except ValueError:
        raise ValueError(f"Wrong pred_rot type: {rot_type}") from None  # model_utils.py:39
    d = hip.ROT_DIMS[type_id]
    if type_id == hip.ROT_QUAT:
        assert rot.ndim == 2 and rot.shape[1] == 4, rot.shape  # pose_utils.py:357
    elif rot.shape[-1] != d:
        raise ValueError(f"Input size must be a (*, {d}) tensor. Got {tuple(rot.shape)}")  # lie_algebra.py:23
    return _RotToMat.apply(rot.reshape(-1, d), type_id)


def _build_head(cfg, head_cfg, num_classes):
    init_cfg = copy.deepcopy(head_cfg.INIT_CFG)
    init_cfg = dict(init_cfg)
    head_type = init_cfg.pop("type")
    init_cfg.update(num_classes=num_classes)
    head = HEADS[head_type](**init_cfg)
    params_lr_list = []
    if cfg.MODEL.CATRE.get("FREEZE_UNUSED_NORM", False):
        # opt-in (not a reference flag): the `norm` GroupNorm no forward uses (conv_out_per_rot_head.py:92,
        # fc_trans_size_head.py:28) stops being a trainable parameter - it stays in the state_dict.  Under the reference's
        # DistributedDataParallel(find_unused_parameters=True) wrap (main_catre.py:154-160) every never-used trainable tensor
        # makes the reducer wait for its used-parameter bitmap and copy it to the host at the end of EVERY backward
        # (a device synchronisation: +1.1 ms per B=256 iteration, bench.py `ddp_world1`).  What changes for a caller: these
        # six tensors are no longer in the optimizer's param groups (the reference lists them; they never receive a
        # gradient there either), so an optimizer state_dict saved by the reference does not load into this optimizer.
        for n, p in head.named_parameters():
            if n.split(".")[-2:-1] == ["norm"]:
                p.requires_grad = False
    if head_cfg.get("FREEZE", False):
        for p in head.parameters():
            p.requires_grad = False
    else:
        params_lr_list.append(
            {
                "params": filter(lambda p: p.requires_grad, head.parameters()),
                "lr": float(cfg.SOLVER.BASE_LR) * head_cfg.get("LR_MULT", 1.0),
            }
        )
    return head, params_lr_list


def get_rot_head(cfg):
    net_cfg = cfg.MODEL.CATRE
    rot_num_classes = net_cfg.NUM_CLASSES if net_cfg.ROT_HEAD.get("CLASS_AWARE", False) else 1
    return _build_head(cfg, net_cfg.ROT_HEAD, rot_num_classes)


def get_ts_head(cfg):
    net_cfg = cfg.MODEL.CATRE
    num_classes = net_cfg.NUM_CLASSES if net_cfg.ROT_HEAD.get("CLASS_AWARE", False) else 1
    return _build_head(cfg, net_cfg.TS_HEAD, num_classes)
