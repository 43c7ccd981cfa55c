"""Device-side training loss: the terms of ``CATRE_disR_shared.catre_loss``
(reference ``core/catre/models/CATRE_disR_shared.py:168-288``) and ``PyPMLoss``
(``core/catre/losses/pm_loss.py:85-194``) for the shipped loss configuration.

The reference picks the symmetry-equivalent ground-truth rotation per object in a Python/numpy loop on the
host (``core/utils/pose_utils.py:472-528``: up to 314 ``re()`` evaluations per symmetric object and a
device->host copy per refine iteration).  Here the candidates of the whole batch are scored at once on the
device and no value ever leaves it, so a training step has no host synchronisation.

These are O(B) / O(B*M*3) reductions on tensors that already live on the GPU; they are written with torch
tensor ops (the model outputs they consume come from the HIP kernels and stay autograd-connected).
"""
import numpy as np
import torch
import torch.nn.functional as F

_sym_cache = {}


def _sym_tensor(sym_infos, device, dtype):
    """list of [S_i,3,3] arrays or None -> (cands [B,Smax+1,3,3] with identity first / as padding, valid mask)."""
    key = (tuple(id(s) if s is not None else None for s in sym_infos), str(device))
    hit = _sym_cache.get(key)
    if hit is not None:
        return hit
    B = len(sym_infos)
    smax = max([0] + [np.asarray(s).reshape(-1, 3, 3).shape[0] for s in sym_infos if s is not None])
    cands = np.tile(np.eye(3, dtype=np.float32), (B, smax + 1, 1, 1))
    valid = np.zeros((B, smax + 1), dtype=bool)
    valid[:, 0] = True
    for i, s in enumerate(sym_infos):
        if s is None:
            continue
        s = np.asarray(s, dtype=np.float32).reshape(-1, 3, 3)
        cands[i, 1:1 + s.shape[0]] = s
        valid[i, 1:1 + s.shape[0]] = True
    out = (torch.from_numpy(cands).to(device=device, dtype=dtype), torch.from_numpy(valid).to(device))
    if len(_sym_cache) > 64:
        _sym_cache.clear()
    _sym_cache[key] = out
    return out


@torch.no_grad()
def get_closest_rot_batch(pred_rots, gt_rots, sym_infos):
    """Batched ``get_closest_rot_batch``: per object argmin over {R_gt, R_gt @ S_k} of the rotational error to the
    prediction.  ``re`` is a decreasing function of trace(R_pred R_cand^T), so the arg-max of the trace is taken;
    the first maximum wins, like the reference's strict ``<`` scan that starts at the un-rotated ground truth."""
    sym, valid = _sym_tensor(sym_infos, gt_rots.device, gt_rots.dtype)
    # 3x3 products as broadcast multiply-adds: torch.matmul would dispatch thousands of tiny GEMMs to hipBLASLt
    cand = (gt_rots.unsqueeze(1).unsqueeze(-1) * sym.unsqueeze(-3)).sum(-2)   # [B,S+1,3,3] = R_gt @ S_k
    tr = (pred_rots.detach().unsqueeze(1) * cand).sum((-1, -2))         # trace(P C^T) = sum_ij P_ij C_ij
    tr = torch.clamp(0.5 * (torch.clamp(tr, max=3.0) - 1.0), -1.0, 1.0)
    tr = torch.where(valid, tr, torch.full_like(tr, -2.0))
    best = torch.argmax(tr, dim=1)
    return cand[torch.arange(cand.shape[0], device=cand.device), best]


def catre_loss(cfg, out_rot, out_trans, out_scale, gt_rot, gt_trans, gt_scale, obj_kps, sym_info):
    loss_cfg = cfg.MODEL.CATRE.LOSS_CFG
    ld = {}
    if loss_cfg.PM_LW > 0:
        assert (obj_kps is not None) and (gt_trans is not None) and (gt_rot is not None)
        if loss_cfg.PM_LOSS_TYPE.lower() != "l1" or not loss_cfg.PM_R_ONLY or loss_cfg.get("PM_USE_BBOX", False):
            raise NotImplementedError("PM loss: the shipped configuration (L1, R-only) is implemented")
        g = get_closest_rot_batch(out_rot, gt_rot, sym_info) if loss_cfg.PM_LOSS_SYM else gt_rot
        if loss_cfg.PM_WITH_SCALE:
            pe, pt = obj_kps * out_scale.unsqueeze(1), obj_kps * gt_scale.unsqueeze(1)
        else:
            pe = pt = obj_kps
        est = (out_rot.unsqueeze(1) * pe.unsqueeze(-2)).sum(-1)          # R (kps * s) per point, [B,M,3]
        tgt = (g.unsqueeze(1) * pt.unsqueeze(-2)).sum(-1)
        ld["loss_PM_R"] = 3 * F.l1_loss(est, tgt) * loss_cfg.PM_LW
    if loss_cfg.ROT_LW > 0:
        # index lists are built on the host from the python list: no device->host sync (torch.where would force one)
        ns = torch.tensor([i for i, s in enumerate(sym_info) if s is None], dtype=torch.long, device=out_rot.device)
        sy = torch.tensor([i for i, s in enumerate(sym_info) if s is not None], dtype=torch.long, device=out_rot.device)
        if ns.numel() > 0:
            if loss_cfg.ROT_LOSS_TYPE == "angular":
                cos = ((out_rot[ns] * gt_rot[ns]).sum((-1, -2)) - 1) / 2     # trace(R_pred R_gt^T)
                ld["loss_rot"] = ((1 - cos) / 2).mean() * loss_cfg.ROT_LW
            elif loss_cfg.ROT_LOSS_TYPE == "L2":
                ld["loss_rot"] = torch.pow(out_rot[ns] - gt_rot[ns], 2).mean() * loss_cfg.ROT_LW
            else:
                raise ValueError(f"Unknown rot loss type: {loss_cfg.ROT_LOSS_TYPE}")
        if sy.numel() > 0:
            if loss_cfg.ROT_YAXIS_LOSS_TYPE == "L1":
                ld["loss_yaxis_rot"] = F.l1_loss(out_rot[sy][:, :, 1], gt_rot[sy][:, :, 1]) * loss_cfg.ROT_LW
            elif loss_cfg.ROT_YAXIS_LOSS_TYPE == "smoothL1":
                ld["loss_yaxis_rot"] = F.smooth_l1_loss(out_rot[sy][:, :, 1], gt_rot[sy][:, :, 1]) * loss_cfg.ROT_LW
            else:
                raise ValueError(f"Unknown rot yaxis loss type: {loss_cfg.ROT_YAXIS_LOSS_TYPE}")
    if loss_cfg.TRANS_LW > 0:
        fn = {"L1": F.l1_loss, "MSE": F.mse_loss}.get(loss_cfg.TRANS_LOSS_TYPE)
        if fn is None:
            raise ValueError(f"Unknown trans loss type: {loss_cfg.TRANS_LOSS_TYPE}")
        if loss_cfg.TRANS_LOSS_DISENTANGLE:
            ld["loss_trans_xy"] = fn(out_trans[:, :2], gt_trans[:, :2]) * loss_cfg.TRANS_LW
            ld["loss_trans_z"] = fn(out_trans[:, 2], gt_trans[:, 2]) * loss_cfg.TRANS_LW
        else:
            ld["loss_trans_LPnP"] = fn(out_trans, gt_trans) * loss_cfg.TRANS_LW
    if loss_cfg.SCALE_LW > 0:
        assert cfg.MODEL.REFINE_SCLAE
        fn = {"L1": F.l1_loss, "MSE": F.mse_loss}.get(loss_cfg.SCALE_LOSS_TYPE)
        if fn is None:
            raise ValueError(f"Unknown scale loss type: {loss_cfg.SCALE_LOSS_TYPE}")
        ld["loss_scale"] = fn(out_scale, gt_scale) * loss_cfg.SCALE_LW
    return ld
