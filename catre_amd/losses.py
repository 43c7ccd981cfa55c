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


# loss-type names of the reference (CATRE_disR_shared.py:232-285) -> the enums of catre_loss_cfg
_YAXIS_TYPES = {"L1": 0, "smoothL1": 1, "L2": 2, "angular": 3}
_NORM_TYPES = {"L1": 0, "MSE": 1, "L2": 2}


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
            raise ValueError(f"Unknown rot loss type: {lc.ROT_LOSS_TYPE}")  # CATRE_disR_shared.py:228
        if lc.ROT_YAXIS_LOSS_TYPE not in _YAXIS_TYPES:
            raise ValueError(f"Unknown rot yaxis loss type: {lc.ROT_YAXIS_LOSS_TYPE}")  # :243
    c.rot_l2, c.yaxis_smooth = int(lc.ROT_LOSS_TYPE == "L2"), _YAXIS_TYPES.get(lc.ROT_YAXIS_LOSS_TYPE, 0)
    c.trans_on = int(lc.TRANS_LW > 0)
    if c.trans_on and lc.TRANS_LOSS_TYPE not in _NORM_TYPES:
        raise ValueError(f"Unknown trans loss type: {lc.TRANS_LOSS_TYPE}")  # :259,269
    c.trans_mse, c.trans_split = _NORM_TYPES.get(lc.TRANS_LOSS_TYPE, 0), int(bool(lc.TRANS_LOSS_DISENTANGLE))
    c.scale_on = int(lc.SCALE_LW > 0)
    if c.scale_on:
        assert cfg.MODEL.REFINE_SCLAE
        if lc.SCALE_LOSS_TYPE not in _NORM_TYPES:
            raise ValueError(f"Unknown scale loss type: {lc.SCALE_LOSS_TYPE}")  # :283
    c.scale_mse = _NORM_TYPES.get(lc.SCALE_LOSS_TYPE, 0)
    c.pm_lw, c.rot_lw, c.trans_lw, c.scale_lw = float(lc.PM_LW), float(lc.ROT_LW), float(lc.TRANS_LW), float(lc.SCALE_LW)
    return c


class _FusedLoss(torch.autograd.Function):
    """-> (losses [6], vis [N_VIS], prefix [len(terms)]): the six loss terms, the logging scalars, and the running sums
    ((0 + l[t0]) + l[t1]) + ... of the terms the loss dict will hold, in its order (see :class:`_LossTerm`)."""

    @staticmethod
    def forward(ctx, pose, scale, gt_rot, gt_trans, gt_scale, kps, cands, valid, is_sym, lcfg, trans_deltas, terms):
        lib = hip.load()
        B, M, S1 = pose.shape[0], (kps.shape[1] if kps is not None else 0), cands.shape[1]
        dev = pose.device
        best = torch.empty(B, dtype=torch.int32, device=dev)
        counts = torch.empty(2, dtype=torch.int32, device=dev)
        part = torch.empty(B * 8, dtype=torch.float32, device=dev)
        n = len(terms)
        buf = torch.empty(6 + N_VIS + n, dtype=torch.float32, device=dev)
        tarr = (ctypes.c_int32 * max(n, 1))(*terms)
        hip.check(lib.catre_loss_fwd_sums(hip.ptr(pose), hip.ptr(scale), hip.ptr(gt_rot), hip.ptr(gt_trans), hip.ptr(gt_scale),
                                          hip.ptr(kps), hip.ptr(cands), hip.ptr(valid), hip.ptr(is_sym), ctypes.byref(lcfg),
                                          hip.ptr(best), hip.ptr(counts), hip.ptr(part), hip.ptr(buf), hip.ptr(trans_deltas),
                                          tarr, n, hip.ptr(buf[6 + N_VIS:]) if n else None, B, M, S1, hip.stream_ptr(dev)),
                  "catre_loss_fwd_sums")
        ctx.save_for_backward(pose, scale, gt_rot, gt_trans, gt_scale, kps, cands, is_sym, best, counts)
        ctx.lcfg, ctx.dims, ctx.terms = lcfg, (B, M, S1), tuple(terms)
        ctx.set_materialize_grads(False)
        losses, vis = buf[:6], buf[6:6 + N_VIS]
        ctx.mark_non_differentiable(vis)
        # the running sums as separate 0-dim outputs: the gradient of the one that gets used arrives alone (an unbind of
        # one vector would zero-fill the other five and stack them)
        return (losses, vis) + tuple(buf[6 + N_VIS + k] for k in range(n))

    @staticmethod
    def backward(ctx, up, _up_vis, *up_sums):
        if up is None and all(u is None for u in up_sums):
            return (None,) * 12
        pose, scale, gt_rot, gt_trans, gt_scale, kps, cands, is_sym, best, counts = ctx.saved_tensors
        B, M, S1 = ctx.dims
        lib = hip.load()
        up = up.contiguous() if up is not None else None
        n = len(ctx.terms)
        tarr = (ctypes.c_int32 * max(n, 1))(*ctx.terms)
        ups = [u.reshape(1).float().contiguous() if u is not None else None for u in up_sums]   # (no-ops for fp32 scalars)
        parr = (ctypes.c_void_p * max(n, 1))(*[u.data_ptr() if u is not None else None for u in ups])
        dpose, dscale = torch.empty_like(pose), torch.empty_like(scale)
        hip.check(lib.catre_loss_bwd_sums(hip.ptr(pose), hip.ptr(scale), hip.ptr(gt_rot), hip.ptr(gt_trans), hip.ptr(gt_scale),
                                          hip.ptr(kps), hip.ptr(cands), hip.ptr(is_sym), hip.ptr(best), hip.ptr(counts),
                                          hip.ptr(up), parr if any(u is not None for u in ups) else None, tarr, n,
                                          ctypes.byref(ctx.lcfg), hip.ptr(dpose), hip.ptr(dscale), B, M, S1,
                                          hip.stream_ptr(pose.device)), "catre_loss_bwd_sums")
        return (dpose, dscale) + (None,) * 10


_ADDS = (torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__, torch.add)
_IADDS = (torch.Tensor.add_, torch.Tensor.__iadd__)   # `acc += v` arrives as add_


class _LossTerm(torch.Tensor):
    """A value of the loss dict (or a running sum of its first values).  The reference's train loop adds the dict up with
    python's ``sum(loss_dict.values())`` (engine.py:318): ``0 + v0``, then ``+ v1`` ...  - one device add per term, and
    autograd's per-term bookkeeping on the way back (ten small launches per iteration).  The loss kernels already hold every
    intermediate of that chain (``prefix[k]``, the same additions in the same order: same bits), so an add that continues
    the chain - ``0 + v0``, ``prefix[k] + v(k+1)``, either operand order, also written ``acc += v`` - returns the precomputed
    tensor, attached to the same autograd node.  Every other use of these tensors is plain torch."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if (func in _ADDS or func in _IADDS) and len(args) == 2 and not kwargs:
            hit = _chain_next(args[0], args[1])
            if hit is None:
                hit = _chain_next(args[1], args[0])
            if hit is not None:
                return hit
        if func in _IADDS and isinstance(args[0], _LossTerm):
            # `acc += v` off the chain, acc being one of these tensors (the loop form of the sum leaves a running sum in
            # acc): they are views of the loss node's output buffer, which autograd does not let anybody modify in place -
            # the statement rebinds acc anyway, so the out-of-place sum is what it gets
            func = torch.Tensor.add
        # everything else: plain torch on plain tensors (aliases on the same autograd graph) - code that asks
        # `type(t) is Tensor` (Tensor.__format__ behind f"{loss:.4f}") sees what it saw before
        with torch._C.DisableTorchFunctionSubclass():
            return func(*_plain(args), **_plain(kwargs or {}))


def _plain(x):
    if isinstance(x, _LossTerm):
        return x.as_subclass(torch.Tensor)
    if isinstance(x, (list, tuple)):
        return type(x)(_plain(v) for v in x)
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    return x


def _chain_next(acc, term):
    """acc + term when that is the next step of the dict's sum chain, else None.  (A term holds the running sum it completes
    and the chain's token; a running sum holds the token and its position - no reference cycles.)"""
    if not isinstance(term, _LossTerm):
        return None
    nxt, pos = getattr(term, "_next_sum", None), getattr(term, "_term_pos", None)
    if nxt is None or pos is None:
        return None
    if pos == 0:
        ok = isinstance(acc, (int, float)) and not isinstance(acc, bool) and acc == 0
    else:
        ok = (isinstance(acc, _LossTerm) and getattr(acc, "_chain_tok", None) is term._chain_tok
              and getattr(acc, "_sum_pos", None) == pos - 1)
    return nxt if ok else None


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


N_VIS = 14
VIS_KEYS = ("error_R", "error_t", "error_tx", "error_ty", "error_tz", "tx_pred", "ty_pred", "tz_pred", "tx_delta", "ty_delta",
            "tz_delta", "tx_gt", "ty_gt", "tz_gt")


class VisScalars:
    """The per-iteration logging scalars of the reference's training forward (``CATRE_disR_shared.py:127-164``: mean
    rotation error [deg], mean translation error [cm], object 0's translation / deltas / ground truth) as ONE device
    tensor written by the loss kernels.  Nothing is copied until a value is asked for: ``as_dict()`` does a single
    14-float device->host copy and returns the reference's ``vis/<name>_<cur_iter>`` keys."""

    def __init__(self, tensor, cur_iter):
        self.tensor, self.cur_iter = tensor, cur_iter
        self._host = None

    def tolist(self):
        if self._host is None:
            self._host = self.tensor.detach().tolist()  # the only host sync
        return self._host

    def as_dict(self):
        return {f"vis/{k}_{self.cur_iter}": v for k, v in zip(VIS_KEYS, self.tolist())}


def catre_loss(cfg, out_rot, out_trans, out_scale, gt_rot, gt_trans, gt_scale, obj_kps, sym_info, trans_deltas=None,
               return_vis=False, pose=None):
    """-> the reference's loss dict (same keys, same values).  Gradients flow to out_rot / out_trans / out_scale.
    ``return_vis``: also return the 14 logging scalars (device tensor, see :class:`VisScalars`).
    ``pose``: the [B,3,4] tensor out_rot / out_trans are the slices of (the model's own call): the kernels take it as it is
    instead of a cat of the two slices, and its gradient arrives whole instead of through two zero-fill + copy + add chains."""
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
    if pose is None:
        pose = torch.cat([out_rot, out_trans.unsqueeze(-1)], -1).contiguous()
    else:
        assert tuple(pose.shape) == (B, 3, 4)
        pose = pose.contiguous()
    f32 = lambda t: hip.require_dev_f32(t.contiguous(), "loss input") if t is not None else None
    gs = f32(gt_scale) if gt_scale is not None else torch.zeros(B, 3, dtype=torch.float32, device=dev)
    td = f32(trans_deltas.detach()) if trans_deltas is not None else None
    # the dict's keys -> loss indices, in the reference's insertion order (CATRE_disR_shared.py:168-288)
    keys = []
    if lcfg.pm_on:
        keys.append(("loss_PM_R", 0))
    if lcfg.rot_on:
        if n_nonsym > 0:
            keys.append(("loss_rot", 1))
        if n_sym > 0:
            keys.append(("loss_yaxis_rot", 2))
    if lcfg.trans_on:
        keys += [("loss_trans_xy", 3), ("loss_trans_z", 4)] if lcfg.trans_split else [("loss_trans_LPnP", 3)]
    if lcfg.scale_on:
        keys.append(("loss_scale", 5))
    losses, vis, *prefix = _FusedLoss.apply(hip.require_dev_f32(pose, "pose"), f32(out_scale), f32(gt_rot), f32(gt_trans), gs,
                                           f32(obj_kps), cands, valid, is_sym, lcfg, td, [i for _, i in keys])
    losses = losses.unbind(0)  # six 0-dim views; their backward is one stack instead of six zero-fill + index + add chains
    tok = object()
    sums = [t.as_subclass(_LossTerm) for t in prefix]
    for k, t in enumerate(sums):
        t._chain_tok, t._sum_pos = tok, k
    ld = {}
    for k, (name, i) in enumerate(keys):
        t = losses[i].as_subclass(_LossTerm)
        t._chain_tok, t._term_pos, t._next_sum = tok, k, sums[k]
        ld[name] = t
    return (ld, vis) if return_vis else ld
