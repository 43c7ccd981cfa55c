"""Device-side training loss: the terms of ``CATRE_disR_shared.catre_loss``
(reference ``core/catre/models/CATRE_disR_shared.py:168-288``) and ``PyPMLoss``
(``core/catre/losses/pm_loss.py:85-194``) for the shipped loss configuration.

The reference picks the symmetry-equivalent ground-truth rotation per object in a Python/numpy loop on the
host (``core/utils/pose_utils.py:472-528``: up to 314 ``re()`` evaluations per symmetric object and a
device->host copy per refine iteration) and then evaluates the terms with ~100 small torch kernels.  Here the
whole loss is two HIP launches forward (``catre_loss_fwd``: candidate arg-max, the point-matching sum and the
per-object terms in one workgroup per object, then an ordered reduction) and one backward (``catre_loss_bwd``),
chained into autograd by :class:`_FusedLoss`; no value ever leaves the device.
"""
import ctypes

import numpy as np
import torch

from . import hip

_sym_cache = {}
_KEYS = ("loss_PM_R", "loss_rot", "loss_yaxis_rot", "loss_trans_xy", "loss_trans_z", "loss_scale")


def _sym_tensor(sym_infos, device, dtype):
    """list of [S_i,3,3] arrays or None -> (cands [B,Smax+1,3,3] with identity first / as padding, valid mask,
    is_sym [B] int32)."""
    key = (tuple(id(s) if s is not None else None for s in sym_infos), str(device))
    hit = _sym_cache.get(key)
    if hit is not None:
        return hit[0]
    B = len(sym_infos)
    smax = max([0] + [np.asarray(s).reshape(-1, 3, 3).shape[0] for s in sym_infos if s is not None])
    cands = np.tile(np.eye(3, dtype=np.float32), (B, smax + 1, 1, 1))
    valid = np.zeros((B, smax + 1), dtype=np.uint8)
    valid[:, 0] = 1
    for i, s in enumerate(sym_infos):
        if s is None:
            continue
        s = np.asarray(s, dtype=np.float32).reshape(-1, 3, 3)
        cands[i, 1:1 + s.shape[0]] = s
        valid[i, 1:1 + s.shape[0]] = 1
    is_sym = np.array([0 if s is None else 1 for s in sym_infos], dtype=np.int32)
    out = (torch.from_numpy(cands).to(device=device, dtype=dtype), torch.from_numpy(valid).to(device),
           torch.from_numpy(is_sym).to(device))
    if len(_sym_cache) > 64:
        _sym_cache.clear()
    _sym_cache[key] = (out, list(sym_infos))  # the arrays stay referenced, so their ids cannot be recycled while cached
    return out


def _loss_cfg_struct(cfg):
    lc = cfg.MODEL.CATRE.LOSS_CFG
    c = hip.CatreLossCfg()
    c.pm_on = int(lc.PM_LW > 0)
    if c.pm_on and (lc.PM_LOSS_TYPE.lower() != "l1" or not lc.PM_R_ONLY or lc.get("PM_USE_BBOX", False)):
        raise NotImplementedError("PM loss: the shipped configuration (L1, R-only) is implemented")
    c.pm_sym, c.pm_with_scale = int(bool(lc.PM_LOSS_SYM)), int(bool(lc.PM_WITH_SCALE))
    c.rot_on = int(lc.ROT_LW > 0)
    if c.rot_on:
        if lc.ROT_LOSS_TYPE not in ("angular", "L2"):
            raise ValueError(f"Unknown rot loss type: {lc.ROT_LOSS_TYPE}")
        if lc.ROT_YAXIS_LOSS_TYPE not in ("L1", "smoothL1"):
            raise ValueError(f"Unknown rot yaxis loss type: {lc.ROT_YAXIS_LOSS_TYPE}")
    c.rot_l2, c.yaxis_smooth = int(lc.ROT_LOSS_TYPE == "L2"), int(lc.ROT_YAXIS_LOSS_TYPE == "smoothL1")
    c.trans_on = int(lc.TRANS_LW > 0)
    if c.trans_on and lc.TRANS_LOSS_TYPE not in ("L1", "MSE"):
        raise ValueError(f"Unknown trans loss type: {lc.TRANS_LOSS_TYPE}")
    c.trans_mse, c.trans_split = int(lc.TRANS_LOSS_TYPE == "MSE"), int(bool(lc.TRANS_LOSS_DISENTANGLE))
    c.scale_on = int(lc.SCALE_LW > 0)
    if c.scale_on:
        assert cfg.MODEL.REFINE_SCLAE
        if lc.SCALE_LOSS_TYPE not in ("L1", "MSE"):
            raise ValueError(f"Unknown scale loss type: {lc.SCALE_LOSS_TYPE}")
    c.scale_mse = int(lc.SCALE_LOSS_TYPE == "MSE")
    c.pm_lw, c.rot_lw, c.trans_lw, c.scale_lw = float(lc.PM_LW), float(lc.ROT_LW), float(lc.TRANS_LW), float(lc.SCALE_LW)
    return c


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, scale, gt_rot, gt_trans, gt_scale, kps, cands, valid, is_sym, lcfg):
        lib = hip.load()
        B, M, S1 = pose.shape[0], (kps.shape[1] if kps is not None else 0), cands.shape[1]
        dev = pose.device
        best = torch.empty(B, dtype=torch.int32, device=dev)
        counts = torch.empty(2, dtype=torch.int32, device=dev)
        part = torch.empty(B * 8, dtype=torch.float32, device=dev)
        losses = torch.empty(6, dtype=torch.float32, device=dev)
        hip.check(lib.catre_loss_fwd(hip.ptr(pose), hip.ptr(scale), hip.ptr(gt_rot), hip.ptr(gt_trans), hip.ptr(gt_scale),
                                     hip.ptr(kps), hip.ptr(cands), hip.ptr(valid), hip.ptr(is_sym), ctypes.byref(lcfg),
                                     hip.ptr(best), hip.ptr(counts), hip.ptr(part), hip.ptr(losses), B, M, S1,
                                     hip.stream_ptr(dev)), "catre_loss_fwd")
        ctx.save_for_backward(pose, scale, gt_rot, gt_trans, gt_scale, kps, cands, is_sym, best, counts)
        ctx.lcfg, ctx.dims = lcfg, (B, M, S1)
        return losses

    @staticmethod
    def backward(ctx, up):
        pose, scale, gt_rot, gt_trans, gt_scale, kps, cands, is_sym, best, counts = ctx.saved_tensors
        B, M, S1 = ctx.dims
        lib = hip.load()
        up = up.contiguous()
        dpose, dscale = torch.empty_like(pose), torch.empty_like(scale)
        hip.check(lib.catre_loss_bwd(hip.ptr(pose), hip.ptr(scale), hip.ptr(gt_rot), hip.ptr(gt_trans), hip.ptr(gt_scale),
                                     hip.ptr(kps), hip.ptr(cands), hip.ptr(is_sym), hip.ptr(best), hip.ptr(counts), hip.ptr(up),
                                     ctypes.byref(ctx.lcfg), hip.ptr(dpose), hip.ptr(dscale), B, M, S1,
                                     hip.stream_ptr(pose.device)), "catre_loss_bwd")
        return dpose, dscale, None, None, None, None, None, None, None, None


class SymTensors:
    """Symmetry info already on the device (what :func:`catre_loss` builds from the python list): ``cands``
    [B,S1,3,3] with the identity first, ``valid`` [B,S1] uint8, ``is_sym`` [B] int32.  Passing this instead of the list
    keeps the call free of host->device copies (HIP-graph capture); both rotation terms are then always reported
    (a term without objects is 0)."""

    def __init__(self, cands, valid, is_sym):
        self.cands, self.valid, self.is_sym = cands, valid, is_sym

    @classmethod
    def from_list(cls, sym_infos, device, s1=None):
        cands, valid, is_sym = _sym_tensor(list(sym_infos), device, torch.float32)
        if s1 is not None and cands.shape[1] < s1:  # pad to a fixed candidate count
            pad = s1 - cands.shape[1]
            eye = torch.eye(3, device=device).expand(cands.shape[0], pad, 3, 3)
            cands = torch.cat([cands, eye], 1).contiguous()
            valid = torch.cat([valid, torch.zeros(valid.shape[0], pad, dtype=valid.dtype, device=device)], 1).contiguous()
        return cls(cands, valid, is_sym)


def catre_loss(cfg, out_rot, out_trans, out_scale, gt_rot, gt_trans, gt_scale, obj_kps, sym_info):
    """-> the reference's loss dict (same keys, same values).  Gradients flow to out_rot / out_trans / out_scale."""
    lc = cfg.MODEL.CATRE.LOSS_CFG
    B = out_rot.shape[0]
    dev = out_rot.device
    if lc.PM_LW > 0:
        assert (obj_kps is not None) and (gt_trans is not None) and (gt_rot is not None)
    if isinstance(sym_info, SymTensors):
        cands, valid, is_sym = sym_info.cands, sym_info.valid, sym_info.is_sym
        n_sym = n_nonsym = 1  # unknown on the host: report both rotation terms
    else:
        sym_info = list(sym_info) if sym_info is not None else [None] * B
        cands, valid, is_sym = _sym_tensor(sym_info, dev, torch.float32)
        n_sym = sum(1 for s in sym_info if s is not None)
        n_nonsym = B - n_sym
    lcfg = _loss_cfg_struct(cfg)
    pose = torch.cat([out_rot, out_trans.unsqueeze(-1)], -1).contiguous()
    f32 = lambda t: hip.require_dev_f32(t.contiguous(), "loss input") if t is not None else None
    gs = f32(gt_scale) if gt_scale is not None else torch.zeros(B, 3, dtype=torch.float32, device=dev)
    losses = _FusedLoss.apply(hip.require_dev_f32(pose, "pose"), f32(out_scale), f32(gt_rot), f32(gt_trans), gs,
                              f32(obj_kps), cands, valid, is_sym, lcfg)
    ld = {}
    if lcfg.pm_on:
        ld["loss_PM_R"] = losses[0]
    if lcfg.rot_on:
        if n_nonsym > 0:
            ld["loss_rot"] = losses[1]
        if n_sym > 0:
            ld["loss_yaxis_rot"] = losses[2]
    if lcfg.trans_on:
        if lcfg.trans_split:
            ld["loss_trans_xy"], ld["loss_trans_z"] = losses[3], losses[4]
        else:
            ld["loss_trans_LPnP"] = losses[3]
    if lcfg.scale_on:
        ld["loss_scale"] = losses[5]
    return ld
