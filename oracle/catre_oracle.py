"""ORACLE - CPU restatement of CATRE's refine hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file.  The product (``catre_amd``) never does: its forward is the HIP library
and it raises when that library is missing.

What this is: a literal, as-written restatement (plain ``torch`` CPU ops, fp32 by default,
fp64 on request) of the reference functions on the path, each citing the reference
``file:line`` it follows (paths relative to ``/root/reference``).  It materialises every
intermediate exactly like the reference does (``repeat``/``cat`` of the global feature
included), so it is slow and only meant for small batches.

Parity pin: the reference holds no tests or golden vectors for this path (SURVEY.md
section 4), so the oracle is pinned against outputs of the reference itself, imported in
the build container by ``oracle/make_golden.py`` and committed as ``tests/golden/*.npz``
(``tests/test_oracle_golden.py`` checks oracle == reference outputs).

All functions take ``sd``: a ``state_dict``-style mapping with the reference's parameter
names (``pcl_net.stn.conv1.weight`` ... see SURVEY.md section 8b).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- reduced-precision emulation
# The reference reaches reduced precision through torch.cuda.amp.autocast (engine.py:304, TEST.AMP_TEST).  The HIP
# bf16 path rounds only the OPERANDS of the per-point GEMMs to bf16 and keeps fp32 accumulation, bias, ReLU, max,
# GroupNorm statistics, GELU, FC tails, ts head and the pose update (catre_amd/csrc/catre_bf16.h).  Inside
# ``with operand_rounding("bf16"):`` the functions below apply exactly that rounding (RNE) at exactly those places,
# so the bf16 kernels can be checked to fp32-re-association accuracy instead of only to bf16 accuracy.
_ROUND = {"mode": None}


class operand_rounding:
    """``"bf16"``: the inference rounding described above.  ``"bf16_train"``: the same forward AND what the autocast
    TRAINING path (engine.py:304,333-347 on ``catre_amd/train_ops.py``) does on the way back, for ``torch.autograd`` through
    these functions: the rounding of an operand is straight-through (the gradient passes unchanged - ``Tensor.to`` already
    differentiates like that), every bf16 GEMM's incoming gradient is rounded to bf16 before it feeds the dgrad and wgrad
    products (their operands are bf16: dY, W, X) while the bias gradient sums the unrounded one, and the rotation heads'
    y0 / y1 rows are bf16 values whose GroupNorm statistics are those of the rounded values (``_RotHeadLP``).
    ``"bf16_train_layerwise"``: the rounding points of the LAYER-WISE autocast ops (shapes off the 64-point grid,
    ``train_forward.pointnet_rows`` / ``_rot_head``) on matrices the tiled bf16 row GEMM takes (>= 2048 rows): every row
    GEMM still rounds its operands and its incoming gradient, but the rows BETWEEN the GEMMs are fp32 - relu(conv1) and the
    feature transform (``k_cloud_matmul64``: an fp32 kernel on the unrounded h1 and T64) are not rounded, max over the
    points of pointfeat sees the unrounded values, and the rotation heads' GroupNorms normalise the unrounded fp32 rows y0 /
    y1 with their own statistics.
    ``"bf16_train_wgrad"``: the same layer-wise ops BELOW 2048 rows (``train_ops._tiled_gemm_ok``; the ``train_b4`` fixture:
    896 rows): forward and data gradients of every layer run on the fp32 small-matrix kernels (``catre_linear`` /
    ``catre_linear_t``) and ONLY the weight gradients are bf16-operand products (``k_gemm_tn_lp``: dY and X rounded, fp32
    accumulation) - except the pooled layers' (``catre_op_maxlin_bwd_w``) and the 3 -> 64 conv1s' (``k_skinny_bwd``), which
    are fp32 kernels in every mode."""

    def __init__(self, mode):
        assert mode in (None, "bf16", "bf16_train", "bf16_train_layerwise", "bf16_train_wgrad")
        self.mode = mode

    def __enter__(self):
        self.prev, _ROUND["mode"] = _ROUND["mode"], self.mode

    def __exit__(self, *a):
        _ROUND["mode"] = self.prev


def _q(t):
    return t.to(torch.bfloat16).to(t.dtype) if _ROUND["mode"] not in (None, "bf16_train_wgrad") else t


# Teacher forcing (tests that compare GRADIENTS at sizes where a rounding emulation is chaotic): the discrete decisions of a
# forward - which units a ReLU lets through, which point wins a max-pool - taken from a recorded run instead of from this
# restatement's own values.  ``with teacher_forcing({key: tensor}):`` keys "<tag><prefix>.<layer>" -> a {0,1} mask shaped like
# the activation [B,C,n], or "<...>.pool" -> the winning point index [B,C] (int64) (+ ".poolrelu": mask [B,C] of the pooled
# ReLU); tags "x." (observed cloud) / "k." (prior).  Autograd then differentiates along the recorded activation pattern.
# ``teacher_forcing(d, record=True)`` fills ``d`` with this run's own decisions instead.
_FORCE = {"d": None, "rec": None}


class teacher_forcing:
    def __init__(self, d, record=False):
        self.d, self.record = d, record

    def __enter__(self):
        self.prev = dict(_FORCE)
        _FORCE["d"], _FORCE["rec"] = (None, self.d) if self.record else (self.d, None)

    def __exit__(self, *a):
        _FORCE.update(self.prev)


def _relu(h, key=None):
    d = _FORCE["d"]
    if d is not None and key in d:
        return h * d[key].to(h.dtype)
    if _FORCE["rec"] is not None and key is not None:
        _FORCE["rec"][key] = (h.detach() > 0)
    return F.relu(h)


def _maxpool(h, key=None):
    """max over the points (dim 2) -> [B,C]"""
    d = _FORCE["d"]
    if d is not None and key in d:
        return torch.gather(h, 2, d[key].unsqueeze(-1)).squeeze(-1)
    m, i = torch.max(h, 2)
    if _FORCE["rec"] is not None and key is not None:
        _FORCE["rec"][key] = i.detach()
    return m


def _train_mode():
    return _ROUND["mode"] in ("bf16_train", "bf16_train_layerwise", "bf16_train_wgrad")


def _q_rows(t):
    """rounding of an activation that the FUSED kernels hold as a bf16 image / bf16 rows; the layer-wise ops keep it fp32"""
    return t if _ROUND["mode"] in ("bf16_train_layerwise", "bf16_train_wgrad") else _q(t)


class _RoundGrad(torch.autograd.Function):
    """identity whose gradient is rounded to bf16 (the dY operand of a bf16 dgrad / wgrad GEMM)"""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _FcWgradRounded(torch.autograd.Function):
    """``F.linear`` whose WEIGHT gradient is a bf16-operand product (dY and X rounded, fp32 accumulation) while the forward,
    the data gradient and the bias gradient stay fp32: the small FC layers of the path under autocast training
    (``train_ops._Linear``: ``catre_linear`` / ``catre_linear_t`` are fp32 kernels, the weight gradient goes through
    ``k_gemm_tn_lp``)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        q = lambda t: t.to(torch.bfloat16).to(t.dtype)
        return dy @ w, q(dy).t() @ q(x), (dy.sum(0) if ctx.has_b else None)


def _fc(x, w, b=None):
    """One FC layer of the path (STN tails, ts head, the global half of rot-head layer 0)."""
    if _train_mode():
        return _FcWgradRounded.apply(x, w, b)
    return F.linear(x, w, b)


class _ConvWgradRounded(torch.autograd.Function):
    """k = 1 ``F.conv1d`` whose WEIGHT gradient alone is a bf16-operand product (:class:`_FcWgradRounded` on [B,C,n] rows)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return F.conv1d(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        q = lambda t: t.to(torch.bfloat16).to(t.dtype)
        dx = torch.einsum("bjn,jc->bcn", dy, w[:, :, 0])
        dw = torch.einsum("bjn,bcn->jc", q(dy), q(x)).unsqueeze(-1)
        return dx, dw, (dy.sum((0, 2)) if ctx.has_b else None)


def _mm(fn, x, w, b=None, pooled=False):
    """One per-point GEMM of the path (``fn``: ``F.conv1d`` with a k=1 weight) under the active rounding mode.  pooled: the
    layer in front of a max-pool (its weight gradient is an fp32 gather kernel in every mode)."""
    mode = _ROUND["mode"]
    if mode is None:
        return fn(x, w, b)
    if mode == "bf16_train_wgrad":
        return fn(x, w, b) if pooled else _ConvWgradRounded.apply(x, w, b)
    if mode == "bf16":
        return fn(_q(x), _q(w), b)
    y = _RoundGrad.apply(fn(_q(x), _q(w)))
    return y if b is None else y + b.reshape(1, -1, 1)


# ----------------------------------------------------------------------------- a1
def pose_apply(pcl, obj_kps, pose, scale, zero_center=True):
    """``batch_updater_test`` core, ``core/catre/engine/batch_test.py:81-97`` with
    ``transform_normed_pts_batch`` ``lib/pysixd/misc.py:1001-1026``.

    pcl [B,N,3], obj_kps [B,M,3], pose [B,3,4], scale [B,3] -> x [B,3,N], tfd_kps [B,3,M]
    """
    B, M = obj_kps.shape[0], obj_kps.shape[1]
    r_est = pose[:, :3, :3]
    t_est = pose[:, :3, 3:4]
    pts = obj_kps * scale.unsqueeze(1)  # misc.py:1017
    pts = r_est.reshape(B, 1, 3, 3) @ pts.reshape(B, M, 3, 1)  # misc.py:1021
    if not zero_center:
        pts = pts + t_est.reshape(B, 1, 3, 1)  # misc.py:1023-1025
    tfd_kps = pts.squeeze(-1).permute(0, 2, 1)  # batch_test.py:91
    if zero_center:
        x = pcl.permute(0, 2, 1) - t_est.reshape(B, 3, 1)  # batch_test.py:94
    else:
        x = pcl.permute(0, 2, 1)
    return x, tfd_kps


# ----------------------------------------------------------------------------- a2 / a4
def stn(x, sd, prefix, k, tag=""):
    """``STN3d.forward`` / ``STNkd.forward``, ``core/catre/models/pointnets/pointnet.py:24-41,57-78``."""
    w = lambda n: sd[f"{prefix}.{n}"]
    key = f"{tag}{prefix}"
    if k == 3:  # 3 -> 64 runs on the VALU in fp32
        h = _relu(F.conv1d(x, w("conv1.weight"), w("conv1.bias")), f"{key}.conv1")
    else:
        h = _relu(_mm(F.conv1d, x, w("conv1.weight"), w("conv1.bias")), f"{key}.conv1")
    h = _relu(_mm(F.conv1d, h, w("conv2.weight"), w("conv2.bias")), f"{key}.conv2")
    h = _mm(F.conv1d, h, w("conv3.weight"), w("conv3.bias"), pooled=True)
    if (_FORCE["d"] is not None and f"{key}.pool" in _FORCE["d"]) or _FORCE["rec"] is not None:
        h = _relu(_maxpool(h, f"{key}.pool"), f"{key}.poolrelu")  # max_n relu(y) = relu(max_n y): the (recorded) winner, then the ReLU
    else:
        h = torch.max(F.relu(h), 2)[0]  # [B,1024]
    pooled = h
    h = F.relu(_fc(h, w("fc1.weight"), w("fc1.bias")))
    h = F.relu(_fc(h, w("fc2.weight"), w("fc2.bias")))
    h = _fc(h, w("fc3.weight"), w("fc3.bias"))
    h = h + torch.eye(k, dtype=h.dtype).reshape(1, k * k)
    return h.reshape(-1, k, k), pooled


# ----------------------------------------------------------------------------- a3 / a5 / a6
def pointnet_feat(x, sd, prefix="pcl_net", feature_transform=True, global_feat=False, detail=False, tag=""):
    """``PointNetfeat.forward``, ``pointnet.py:97-121``.  x [B,3,n] -> [B,1088,n]."""
    w = lambda n: sd[f"{prefix}.{n}"]
    n_pts = x.shape[2]
    trans, pool3 = stn(x, sd, f"{prefix}.stn", 3, tag)
    h = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)  # :100-102
    h = _q_rows(_relu(F.conv1d(h, w("conv1.weight"), w("conv1.bias")), f"{tag}{prefix}.conv1"))  # :103
    trans_feat, pool64 = None, None
    if feature_transform:
        trans_feat, pool64 = stn(h, sd, f"{prefix}.fstn", 64, tag)  # :106
        h = _q_rows(torch.bmm(h.transpose(2, 1), _q_rows(trans_feat)).transpose(2, 1))  # :107-109
    pointfeat = h  # :111
    h = _relu(_mm(F.conv1d, h, w("conv2.weight"), w("conv2.bias")), f"{tag}{prefix}.conv2")  # (h is rounded already: _q is idempotent)
    h = _relu(_mm(F.conv1d, h, w("conv3.weight"), w("conv3.bias")), f"{tag}{prefix}.conv3")
    h = _mm(F.conv1d, h, w("conv4.weight"), w("conv4.bias"), pooled=True)  # no ReLU, :114
    g = _maxpool(h, f"{tag}{prefix}.pool")  # :115-116
    if global_feat:
        out = g
    else:
        out = torch.cat([g.unsqueeze(-1).repeat(1, 1, n_pts), pointfeat], 1)  # :120-121
    if detail:
        return out, dict(trans=trans, trans_feat=trans_feat, pointfeat=pointfeat, g=g,
                         stn_pool=pool3, fstn_pool=pool64)
    return out


# ----------------------------------------------------------------------------- a8
def gelu_exact(v):
    """``nn.GELU()`` default = exact erf form (``layer_utils.py:83-84``)."""
    return F.gelu(v)


def ts_head(feat, sd, prefix="ts_head", num_gn_groups=32):
    """``FC_TransSizeHead.forward``, ``heads/fc_trans_size_head.py:61-70`` (layers built ``:33-45``)."""
    w = lambda n: sd[f"{prefix}.{n}"]
    h = _fc(feat, w("linears.0.weight"), w("linears.0.bias"))
    h = F.group_norm(h, num_gn_groups, w("linears.1.weight"), w("linears.1.bias"), 1e-5)
    h = gelu_exact(h)
    h = _fc(h, w("linears.3.weight"), w("linears.3.bias"))
    h = F.group_norm(h, num_gn_groups, w("linears.4.weight"), w("linears.4.bias"), 1e-5)
    h = gelu_exact(h)
    return _fc(h, w("fc_t.weight"), w("fc_t.bias")), _fc(h, w("fc_s.weight"), w("fc_s.bias"))


# ----------------------------------------------------------------------------- a9
def rot_head_single(feat, sd, prefix, num_gn_groups=32):
    """``RotHead.forward``, ``heads/conv_out_per_rot_head.py:126-140``.  feat [B,1088,P] -> [B,3]."""
    w = lambda n: sd[f"{prefix}.{n}"]
    if _ROUND["mode"] is None:
        h = F.conv1d(feat, w("layers.0.weight"), w("layers.0.bias"))
        h = F.group_norm(h, num_gn_groups, w("layers.1.weight"), w("layers.1.bias"), 1e-5)
        h = gelu_exact(h)
        h = F.conv1d(h, w("layers.3.weight"), w("layers.3.bias"))
        h = F.group_norm(h, num_gn_groups, w("layers.4.weight"), w("layers.4.bias"), 1e-5)
    else:
        # global-feature channels (0..1023) go through the fp32 FC kernel, the 64 pointfeat channels (already
        # rounded) through the bf16 GEMM; y1 is stored rounded but normalised with the statistics of the
        # unrounded values
        w0 = w("layers.0.weight")
        if _train_mode():
            # the global half as the product computes it: ONE FC row per cloud (the feature is constant over a cloud's points;
            # feat = cat(observed, prior) along the points), broadcast as a per-cloud bias
            Pn = feat.shape[2]
            diff = (feat[0, :1024] != feat[0, :1024, :1]).any(0)                # first point whose global feature differs
            nobs = int(diff.to(torch.int8).argmax()) if bool(diff.any()) else Pn   # = points of the first (observed) cloud
            gl = torch.stack([feat[:, :1024, 0], feat[:, :1024, Pn - 1]], 1)    # [B, 2, 1024]
            bias = _fc(gl.reshape(-1, 1024), w0[:, :1024, 0], w("layers.0.bias")).reshape(-1, 2, w0.shape[0])
            hg = torch.cat([bias[:, 0, :, None].expand(-1, -1, nobs), bias[:, 1, :, None].expand(-1, -1, Pn - nobs)], 2)
            h = hg + _mm(F.conv1d, feat[:, 1024:], w0[:, 1024:])
        else:
            h = F.conv1d(feat[:, :1024], w0[:, :1024], w("layers.0.bias")) + _mm(F.conv1d, feat[:, 1024:], w0[:, 1024:])
        train = _train_mode()
        # training (_RotHeadLP): y0 / y1 are bf16 rows and GroupNorm's statistics are those of the rounded values (what
        # autocast's GroupNorm sees; the layer-wise ops keep fp32 rows: _q_rows); inference (k_rot_l1_bf): y0 stays fp32 on
        # chip, y1 is stored rounded but normalised with the statistics of the unrounded values
        h = F.group_norm(_q_rows(h) if train else h, num_gn_groups, w("layers.1.weight"), w("layers.1.bias"), 1e-5)
        h = gelu_exact(h)
        h = _mm(F.conv1d, h, w("layers.3.weight"), w("layers.3.bias"))
        if train:
            h = F.group_norm(_q_rows(h), num_gn_groups, w("layers.4.weight"), w("layers.4.bias"), 1e-5)
            h = gelu_exact(h)
            if _ROUND["mode"] == "bf16_train_wgrad":   # the neck as a layer-wise linear (heads.neck_rows): bf16 weight gradient
                h = _ConvWgradRounded.apply(h, w("neck.0.weight"), w("neck.0.bias")).permute(0, 2, 1)
            else:
                h = F.conv1d(h, w("neck.0.weight"), w("neck.0.bias")).permute(0, 2, 1)
            return F.conv1d(h, w("conv_p.weight"), sd.get(f"{prefix}.conv_p.bias")).squeeze(1).contiguous()
        B, C, P = h.shape
        hg = h.reshape(B, num_gn_groups, -1)
        mean, var = hg.mean(-1, keepdim=True), hg.var(-1, unbiased=False, keepdim=True)
        h = ((_q(h).reshape(B, num_gn_groups, -1) - mean) / torch.sqrt(var + 1e-5)).reshape(B, C, P)
        h = h * w("layers.4.weight").reshape(1, C, 1) + w("layers.4.bias").reshape(1, C, 1)
    h = gelu_exact(h)
    h = F.conv1d(h, w("neck.0.weight"), w("neck.0.bias"))  # [B,3,P]
    h = h.permute(0, 2, 1)  # [B,P,3]
    h = F.conv1d(h, w("conv_p.weight"), sd.get(f"{prefix}.conv_p.bias"))  # [B,1,3]
    return h.squeeze(1).contiguous()


def rot_head(feat, sd, prefix="rot_head"):
    """``ConvOutPerRotHead.forward``, ``conv_out_per_rot_head.py:62-71`` (``per_rot_sup=False``)."""
    rx = rot_head_single(feat, sd, f"{prefix}.rot_head_x")
    ry = rot_head_single(feat, sd, f"{prefix}.rot_head_y")
    return torch.cat((rx, ry), dim=1)


# ----------------------------------------------------------------------------- a10
def rot6d_to_mat_batch(d6):
    """``core/utils/rot_reps.py:34-55``."""
    x_raw, y_raw = d6[..., 0:3], d6[..., 3:6]
    x = F.normalize(x_raw, p=2, dim=-1)
    z = torch.cross(x, y_raw, dim=-1)
    z = F.normalize(z, p=2, dim=-1)
    y = torch.cross(z, x, dim=-1)
    return torch.stack((x, y, z), dim=-1)


def quat2mat_torch(quat, eps=0.0):
    """``core/utils/pose_utils.py:349-412`` (w,x,y,z; normalised by ``norm + eps``)."""
    assert quat.ndim == 2 and quat.shape[1] == 4, quat.shape
    norm_quat = quat.norm(p=2, dim=1, keepdim=True)
    norm_quat = quat / (norm_quat + eps)
    qw, qx, qy, qz = norm_quat[:, 0], norm_quat[:, 1], norm_quat[:, 2], norm_quat[:, 3]
    B = quat.size(0)
    s = 2.0
    X, Y, Z = qx * s, qy * s, qz * s
    wX, wY, wZ = qw * X, qw * Y, qw * Z
    xX, xY, xZ = qx * X, qx * Y, qx * Z
    yY, yZ, zZ = qy * Y, qy * Z, qz * Z
    return torch.stack(
        [1.0 - (yY + zZ), xY - wZ, xZ + wY, xY + wZ, 1.0 - (xX + zZ), yZ - wX, xZ - wY, yZ + wX, 1.0 - (xX + yY)],
        dim=1,
    ).reshape(B, 3, 3)


def allo_to_ego_mat_torch(translation, rot_allo, eps=1e-4):
    """``core/utils/utils.py:200-231``."""
    cam_ray = torch.tensor([0, 0, 1.0], dtype=translation.dtype)
    obj_ray = translation / (torch.norm(translation, dim=1, keepdim=True) + eps)
    angle = obj_ray[:, 2:3].acos()
    axis = torch.cross(cam_ray.expand_as(obj_ray), obj_ray, dim=-1)
    axis = axis / (torch.norm(axis, dim=1, keepdim=True) + eps)
    q = torch.cat(
        [
            torch.cos(angle / 2.0),
            axis[:, 0:1] * torch.sin(angle / 2.0),
            axis[:, 1:2] * torch.sin(angle / 2.0),
            axis[:, 2:3] * torch.sin(angle / 2.0),
        ],
        dim=1,
    )
    return torch.matmul(quat2mat_torch(q), rot_allo)


def qexp(q, eps=1e-8):
    """``core/utils/quaternion_lf.py:294-317`` (``is_normalized=False``): exponent of (s; v), or of the pure quaternion
    (0; v) when ``q`` is [B,3]."""
    if q.shape[1] == 4:
        s, v = torch.split(q, (1, 3), dim=-1)
    else:
        s = torch.zeros_like(q[:, :1])
        v = q
    theta = torch.norm(v, dim=-1, keepdim=True)
    exp_s = torch.exp(s)
    w = torch.cos(theta)
    xyz = 1.0 / theta.clamp(min=eps) * torch.sin(theta) * v
    return exp_s * torch.cat((w, xyz), dim=-1)


def lie_vec_to_rot(angle_axis):
    """``core/utils/lie_algebra.py:7-77``: Rodrigues with the axis ``v / (theta + 1e-6)``; first-order matrix where
    ``theta^2 <= 1e-6``."""
    if not angle_axis.shape[-1] == 3:
        raise ValueError("Input size must be a (*, 3) tensor. Got {}".format(angle_axis.shape))
    eps = 1e-6
    theta2 = (angle_axis * angle_axis).sum(1, keepdim=True)  # matmul(v, v^T) (:56-58)
    theta = torch.sqrt(theta2)
    wxyz = angle_axis / (theta + eps)
    wx, wy, wz = torch.chunk(wxyz, 3, dim=1)
    c, sn = torch.cos(theta), torch.sin(theta)
    r00 = c + wx * wx * (1.0 - c)
    r10 = wz * sn + wx * wy * (1.0 - c)
    r20 = -wy * sn + wx * wz * (1.0 - c)
    r01 = wx * wy * (1.0 - c) - wz * sn
    r11 = c + wy * wy * (1.0 - c)
    r21 = wx * sn + wy * wz * (1.0 - c)
    r02 = wy * sn + wx * wz * (1.0 - c)
    r12 = -wx * sn + wy * wz * (1.0 - c)
    r22 = c + wz * wz * (1.0 - c)
    normal = torch.cat([r00, r01, r02, r10, r11, r12, r20, r21, r22], dim=1).view(-1, 3, 3)
    rx, ry, rz = torch.chunk(angle_axis, 3, dim=1)
    one = torch.ones_like(rx)
    taylor = torch.cat([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1).view(-1, 3, 3)
    mask = (theta2 > eps).view(-1, 1, 1)
    return mask.type_as(theta2) * normal + (mask == False).type_as(theta2) * taylor  # noqa: E712


def get_rot_mat(rot, rot_type):
    """``core/catre/models/model_utils.py:28-40``."""
    if rot_type in ["ego_quat", "allo_quat"]:
        return quat2mat_torch(rot)
    if rot_type in ["ego_log_quat", "allo_log_quat"]:
        return quat2mat_torch(qexp(rot))
    if rot_type in ["ego_lie_vec", "allo_lie_vec"]:
        return lie_vec_to_rot(rot)
    if rot_type in ["ego_rot6d", "allo_rot6d"]:
        return rot6d_to_mat_batch(rot)
    raise ValueError(f"Wrong pred_rot type: {rot_type}")


# ----------------------------------------------------------------------------- a11
def pose_scale_from_delta_init(
    rot_deltas, trans_deltas, scale_deltas, rot_inits, trans_inits, scale_inits, Ks=None,
    K_aware=False, delta_T_space="3D", delta_T_weight=1.0, delta_z_style="cosypose",
    eps=1e-4, is_allo=False, scale_type="add_iter",
):
    """``core/catre/models/pose_scale_from_delta_init.py:8-95``."""
    bs = rot_deltas.shape[0]
    assert rot_deltas.shape == (bs, 3, 3) and rot_inits.shape == (bs, 3, 3)
    assert trans_deltas.shape == (bs, 3) and trans_inits.shape == (bs, 3)
    trans_deltas = trans_deltas * delta_T_weight
    if delta_T_space == "image":
        zsrc = trans_inits[:, [2]]
        vz = trans_deltas[:, [2]]
        if delta_z_style == "cosypose":
            ztgt = vz * zsrc
        else:
            ztgt = torch.div(zsrc, torch.exp(vz))
        vxvy = trans_deltas[:, :2]
        if K_aware:
            assert Ks is not None and Ks.shape == (bs, 3, 3)
            fxfy = Ks[:, [0, 1], [0, 1]]
        else:
            fxfy = torch.ones_like(vxvy)
        xy_src = trans_inits[:, :2]
        xy_tgt = ztgt * (vxvy / fxfy + xy_src / zsrc)
        trans_tgts = torch.cat([xy_tgt, ztgt], dim=-1)
    elif delta_T_space == "3D":
        trans_tgts = trans_inits + trans_deltas
    else:
        raise ValueError("Unknown delta_T_space: {}".format(delta_T_space))
    if "add" in scale_type:
        scale_tgts = scale_inits + scale_deltas
    else:
        scale_tgts = scale_inits * torch.exp(scale_deltas)
    ego_rot_deltas = allo_to_ego_mat_torch(trans_tgts, rot_deltas, eps=eps) if is_allo else rot_deltas
    rot_tgts = ego_rot_deltas @ rot_inits
    return rot_tgts, trans_tgts, scale_tgts


# ----------------------------------------------------------------------------- a7 / a12
def model_forward(x, tfd_kps, init_pose, init_scale, sd, cfg, K_zoom=None, mean_scales=None, detail=False):
    """``CATRE_disR_shared.forward`` test path, ``core/catre/models/CATRE_disR_shared.py:57-124``."""
    net_cfg = cfg.MODEL.CATRE
    rh, th = net_cfg.ROT_HEAD, net_cfg.TS_HEAD
    pn = net_cfg.PCLNET.INIT_CFG
    ft = pn.get("feature_transform", False)
    pcl_feat, dx = pointnet_feat(x, sd, "pcl_net", ft, pn.get("global_feat", True), detail=True, tag="x.")  # :66
    kps_feat, dk = pointnet_feat(tfd_kps, sd, "pcl_net", ft, pn.get("global_feat", True), detail=True, tag="k.")  # :67
    flat_pcl_feat = _maxpool(pcl_feat, "x.flat")  # :69
    if th.WITH_KPS_FEATURE:
        ts_feat = torch.cat((flat_pcl_feat, _maxpool(kps_feat, "k.flat")), dim=1)  # :71-73
    else:
        ts_feat = flat_pcl_feat
    if th.WITH_INIT_SCALE:
        ts_feat = torch.cat((ts_feat, init_scale), dim=1)  # :78-79
    if th.get("WITH_INIT_TRANS", False):
        ts_feat = torch.cat((ts_feat, init_pose[:, :3, 3]), dim=1)  # :80-82
    trans_deltas, scale_deltas = ts_head(ts_feat, sd, "ts_head", th.INIT_CFG.get("num_gn_groups", 32))  # :84
    rot_feat = torch.cat((pcl_feat, kps_feat), dim=2)  # :86
    rot_deltas = rot_head(rot_feat, sd, "rot_head")  # :88
    rot_m = get_rot_mat(rot_deltas, rh.ROT_TYPE)  # :98
    R, t, s = pose_scale_from_delta_init(  # :100-115
        rot_m, trans_deltas, scale_deltas, init_pose[:, :3, :3], init_pose[:, :3, 3],
        init_scale if "iter" in rh.SCLAE_TYPE else mean_scales, Ks=K_zoom,
        K_aware=rh.T_TRANSFORM_K_AWARE, delta_T_space=rh.DELTA_T_SPACE, delta_T_weight=rh.DELTA_T_WEIGHT,
        delta_z_style=rh.DELTA_Z_STYLE, eps=1e-4, is_allo="allo" in rh.ROT_TYPE, scale_type=rh.SCLAE_TYPE,
    )
    pose = torch.cat([R, t.reshape(-1, 3, 1)], dim=-1)  # :116
    if not cfg.MODEL.REFINE_SCLAE:
        s = init_scale  # :119-120
    if detail:
        d = dict(
            trans_x=dx["trans"], transfeat_x=dx["trans_feat"], g_x=dx["g"], pointfeat_x=dx["pointfeat"],
            trans_k=dk["trans"], transfeat_k=dk["trans_feat"], g_k=dk["g"], pointfeat_k=dk["pointfeat"],
            stn_pool_x=dx["stn_pool"], fstn_pool_x=dx["fstn_pool"],
            stn_pool_k=dk["stn_pool"], fstn_pool_k=dk["fstn_pool"],
            flat_pcl_feat=flat_pcl_feat, trans_deltas=trans_deltas, scale_deltas=scale_deltas,
            rot_deltas=rot_deltas, rot_m=rot_m,
        )
        return pose, s, d
    return pose, s


# ----------------------------------------------------------------------------- a13
def refine_k(batch, sd, cfg, n_iter=None, detail_iter=None):
    """The K-loop of ``catre_inference_on_dataset``, ``core/catre/engine/catre_evaluator.py:292-311``.

    ``batch``: dict with ``pcl, obj_kps, obj_pose_est, obj_scale_est, K, obj_mean_scales``.
    Returns ``out_dict`` with ``pose_0..K`` / ``scale_0..K`` (+ ``detail`` of iteration ``detail_iter``).
    """
    n_iter = cfg.MODEL.CATRE.N_ITER_TEST if n_iter is None else n_iter
    pose, scale = batch["obj_pose_est"], batch["obj_scale_est"]
    out = {"pose_0": pose, "scale_0": scale}
    for i in range(1, n_iter + 1):
        x, tfd = pose_apply(batch["pcl"], batch["obj_kps"], pose, scale, cfg.INPUT.ZERO_CENTER_INPUT)
        r = model_forward(x, tfd, pose, scale, sd, cfg, K_zoom=batch["K"],
                          mean_scales=batch.get("obj_mean_scales"), detail=(detail_iter == i))
        if detail_iter == i:
            out["detail"] = r[2]
        new_pose, new_scale = r[0], r[1]
        pose = new_pose
        if cfg.MODEL.REFINE_SCLAE:  # batch_test.py:74-75
            scale = new_scale
        out[f"pose_{i}"], out[f"scale_{i}"] = new_pose, new_scale
    return out


def cast_sd(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def colmax(x):
    """Stand-alone channel-wise max-pool ``torch.max(x, 2)[0]`` (``pointnet.py:28,61,115``)."""
    return torch.max(x, 2)[0]


GELU_C = 1.0 / math.sqrt(2.0)


# ----------------------------------------------------------------------------- training loss
def get_closest_rot_batch(pred_rots, gt_rots, sym_infos):
    """``core/utils/pose_utils.py:472-528``: per object, the symmetry-equivalent ground-truth rotation with the
    smallest rotational error ``re`` (``lib/pysixd/pose_error.py:359-374``) to the prediction; strict ``<`` keeps
    the first minimum, starting from the un-rotated ground truth."""
    import numpy as np

    out = gt_rots.clone()
    P = pred_rots.detach().cpu().double().numpy()
    G = gt_rots.detach().cpu().double().numpy()

    def re(a, b):
        tr = min(np.trace(a @ b.T), 3.0)
        return np.rad2deg(np.arccos(min(1.0, max(-1.0, 0.5 * (tr - 1.0)))))

    for i, sym in enumerate(sym_infos):
        if sym is None:
            continue
        sym = np.asarray(sym, dtype=np.float64).reshape(-1, 3, 3)
        best, err = G[i], re(P[i], G[i])
        for k in range(sym.shape[0]):
            cand = G[i] @ sym[k]
            e = re(P[i], cand)
            if e < err:
                err, best = e, cand
        out[i] = torch.as_tensor(best, dtype=gt_rots.dtype)
    return out


def l2_loss(pred, target):
    """``L2Loss(reduction="mean")``, ``core/catre/losses/l2_loss.py:5-28``: mean over the batch of per-sample L2 norms."""
    bs = pred.size(0)
    return torch.norm((pred - target).view(bs, -1), p=2, dim=1, keepdim=True).mean()


def angular_distance_vec(v1, v2):
    """``core/catre/losses/rot_loss.py:33-42``."""
    cos = torch.bmm(v1.unsqueeze(1), v2.unsqueeze(2)).squeeze() / (torch.norm(v1, dim=1) * torch.norm(v2, dim=1))
    return ((1 - cos) / 2).mean()


def catre_loss(out_rot, out_trans, out_scale, gt_rot, gt_trans, gt_scale, obj_kps, sym_info, loss_cfg):
    """``CATRE_disR_shared.catre_loss`` (``core/catre/models/CATRE_disR_shared.py:168-288``) with ``PyPMLoss``
    (``core/catre/losses/pm_loss.py:85-194``) for the shipped loss types (L1 PM / angular rot / L1 y-axis / L1 t,s)."""
    ld = {}
    if loss_cfg.PM_LW > 0:
        assert loss_cfg.PM_LOSS_TYPE.lower() == "l1" and loss_cfg.PM_R_ONLY and loss_cfg.PM_WITH_SCALE
        g = get_closest_rot_batch(out_rot, gt_rot, sym_info) if loss_cfg.PM_LOSS_SYM else gt_rot
        est = (out_rot.unsqueeze(1) @ (obj_kps * out_scale.unsqueeze(1)).unsqueeze(-1)).squeeze(-1)
        tgt = (g.unsqueeze(1) @ (obj_kps * gt_scale.unsqueeze(1)).unsqueeze(-1)).squeeze(-1)
        ld["loss_PM_R"] = 3 * F.l1_loss(est, tgt) * loss_cfg.PM_LW
    if loss_cfg.ROT_LW > 0:
        sym_mask = torch.tensor([0 if s is None else 1 for s in sym_info])
        ns, sy = torch.where(sym_mask == 0)[0], torch.where(sym_mask == 1)[0]
        if len(ns) > 0:
            if loss_cfg.ROT_LOSS_TYPE == "angular":  # angular_distance, core/catre/losses/rot_loss.py
                m = torch.bmm(out_rot[ns], gt_rot[ns].transpose(1, 2))
                cos = (torch.einsum("bii->b", m) - 1) / 2
                ld["loss_rot"] = ((1 - cos) / 2).mean() * loss_cfg.ROT_LW
            else:  # "L2": rot_l2_loss = mean of squared element differences
                assert loss_cfg.ROT_LOSS_TYPE == "L2"
                ld["loss_rot"] = torch.pow(out_rot[ns] - gt_rot[ns], 2).mean() * loss_cfg.ROT_LW
        if len(sy) > 0:
            fn = {"L1": F.l1_loss, "smoothL1": F.smooth_l1_loss, "L2": l2_loss,
                  "angular": angular_distance_vec}[loss_cfg.ROT_YAXIS_LOSS_TYPE]  # :232-243
            ld["loss_yaxis_rot"] = fn(out_rot[sy][:, :, 1], gt_rot[sy][:, :, 1]) * loss_cfg.ROT_LW
    if loss_cfg.TRANS_LW > 0:
        fn = {"L1": F.l1_loss, "MSE": F.mse_loss, "L2": l2_loss}[loss_cfg.TRANS_LOSS_TYPE]
        if loss_cfg.TRANS_LOSS_DISENTANGLE:
            ld["loss_trans_xy"] = fn(out_trans[:, :2], gt_trans[:, :2]) * loss_cfg.TRANS_LW
            ld["loss_trans_z"] = fn(out_trans[:, 2], gt_trans[:, 2]) * loss_cfg.TRANS_LW
        else:
            ld["loss_trans_LPnP"] = fn(out_trans, gt_trans) * loss_cfg.TRANS_LW
    if loss_cfg.SCALE_LW > 0:
        fn = {"L1": F.l1_loss, "MSE": F.mse_loss, "L2": l2_loss}[loss_cfg.SCALE_LOSS_TYPE]
        ld["loss_scale"] = fn(out_scale, gt_scale) * loss_cfg.SCALE_LW
    return ld


def y_axis_symmetries(n):
    """n-1 rotations about the y axis by multiples of 2*pi/n (shape of ``get_axis_symmetry_transformations``,
    ``lib/pysixd/misc.py:220-231``, for the NOCS y-symmetric categories ``ref/nocs.py:138-158``)."""
    import numpy as np

    out = []
    for i in range(1, n):
        a = 2.0 * np.pi * i / n
        out.append([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    return np.asarray(out, dtype=np.float32)
