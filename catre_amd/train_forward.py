"""Autograd-connected forward of one refine iteration for TRAINING (``do_loss=True`` path of the reference's
``CATRE_disR_shared.forward``, ``core/catre/models/CATRE_disR_shared.py:57-124``).

Same arithmetic as the fused inference kernels, but laid out layer by layer on point-major activation
matrices so that ``torch.autograd`` can chain the hand-written HIP backward kernels of
``catre_amd/train_ops.py``.  The restructurings of the inference path are kept (the repeated global
feature becomes a per-cloud bias of rot-head layer 0; the ``[B,1088,N]`` tensors are never built), so the
gradients equal the reference's up to fp32 re-association.
"""
import torch

from . import hip
from . import train_ops as T


def _points_rows(x):
    """[B,3,n] (any strides) -> [B*n, 3] contiguous rows (pure data movement)."""
    return x.permute(0, 2, 1).reshape(-1, 3).contiguous()


def _cloud_major_rows(x, tfd_kps):
    """x [B,3,N], tfd_kps [B,3,M] -> [B*N + B*M, 3] cloud-major point rows.  ``runtime.pose_apply`` (``batch_updater_test``)
    writes the two clouds back to back into one buffer: then the matrix is a view of it; anything else is concatenated."""
    xr, kr = _points_rows(x), _points_rows(tfd_kps)
    if (not xr.requires_grad and not kr.requires_grad and xr.dtype == kr.dtype == torch.float32 and xr.device == kr.device
            and xr.untyped_storage().data_ptr() == kr.untyped_storage().data_ptr()
            and kr.data_ptr() == xr.data_ptr() + xr.numel() * 4):
        return torch.as_strided(xr, (xr.shape[0] + kr.shape[0], 3), (3, 1))
    return torch.cat([xr, kr], 0)


def _stn(rows, p, prefix, k, B, N, M, pre=None):
    """STN3d / STNkd on cloud-major rows [R, k] -> [C, k, k]  (pointnet.py:24-41 / 57-78).  ``pre``: (conv1 out, conv2
    out, pooled maxima, arg-max rows) already computed by the fused forward kernel - the nodes then only build the graph."""
    w = lambda n: p[f"{prefix}.{n}"]
    p1, p2, pg = (pre[0], pre[1], (pre[2], pre[3])) if pre is not None else (None, None, None)
    if pre is not None and T.pooled_chain_ok(rows, w("conv1.weight"), w("conv2.weight"), w("conv3.weight"), N, M):
        # the conv stack + pool as one node with a row-sparse backward (train_ops._PooledChain)
        g = T.pooled_chain(rows, w("conv1.weight"), w("conv1.bias"), w("conv2.weight"), w("conv2.bias"),
                           w("conv3.weight"), w("conv3.bias"), True, B, N, M, pre)
    else:
        h = T.linear(rows, w("conv1.weight"), w("conv1.bias"), relu=True, pre=p1)
        h = T.linear(h, w("conv2.weight"), w("conv2.bias"), relu=True, pre=p2)
        g = T.linear_maxpool(h, w("conv3.weight"), w("conv3.bias"), True, B, N, M, pre=pg)  # relu(conv3) then max
    h = T.linear(g, w("fc1.weight"), w("fc1.bias"), relu=True)
    h = T.linear(h, w("fc2.weight"), w("fc2.bias"), relu=True)
    t = T.linear(h, w("fc3.weight"), w("fc3.bias"), identity_k=k)
    return t.view(-1, k, k)


def pointnet_rows_fused(pts, desc, rt, p, B, N, M, prefix="pcl_net", mode=0, obj_copy=True):
    """The same graph as :func:`pointnet_rows` (feature_transform=True), but the three conv stacks run as the FUSED encoder
    kernels with extra stores (``catre_train_{stn3d,stnkd,trunk}_fwd``): one launch per block instead of a row GEMM per
    layer.  Every layer op becomes a graph node around an output that exists already (``pre=``); the backward is the
    layer-wise one, unchanged.  N, M multiples of 64.  mode 0: the fp32 kernels; mode 2: the split kernels (rows hold hi + lo);
    mode 1 (autocast): the bf16-operand kernels -
    the rows they save behind each stack's first layer are bf16 ROWS (half the bytes; the values the reduced-precision dgrad /
    wgrad kernels would round their operands to anyway), read in place by the row-sparse chains' backward - so mode 1
    wants all three stacks to be chains (`fused_lp_ok`)."""
    w = lambda n: p[f"{prefix}.{n}"]
    dev = pts.device
    # fp32: the STN stacks store no activation rows (their row-sparse backward rebuilds them on its live rows)
    buf = rt.train_encoder_buffers(B, N, M, dev,
                                   stn_rows=not (mode == 0 and T.knobs().stn_recompute and not pts.requires_grad), mode=mode)
    packs = rt.train_stn3d(desc, buf, B, N, M, dev, mode)   # (params, packed image): re-packed here, shared by the three kernels
    trans = _stn(pts, p, f"{prefix}.stn", 3, B, N, M, pre=(buf["a1"], buf["a2"], buf["g_stn"], buf["i_stn"]))
    trans3 = trans.detach().reshape(-1, 9).contiguous()
    # x1 / h1 are written by the trunk kernel further down (same stream, before anything reads them)
    x1 = T.cloud_matmul(pts, trans, B, N, M, out_cols=8, pre=buf["x1"])
    h1, h1s = T.linear_fan2(x1, w("conv1.weight"), w("conv1.bias"), relu=True, pre=buf["h1"])  # two consumers, one backward pass
    rt.train_stnkd(desc, trans3, buf, B, N, M, dev, mode, packs=packs)
    trans_feat = _stn(h1s, p, f"{prefix}.fstn", 64, B, N, M, pre=(buf["f1"], buf["f2"], buf["g_fstn"], buf["i_fstn"]))
    trans64 = trans_feat.detach().reshape(-1, 4096).contiguous()
    rt.train_trunk(desc, trans3, trans64, buf, B, N, M, dev, mode, packs=packs)
    pf = T.cloud_matmul(h1, trans_feat, B, N, M, pre=buf["pf"])
    if T.pooled_chain_ok(pf, w("conv2.weight"), w("conv3.weight"), w("conv4.weight"), N, M):
        # the conv stack with its row-sparse backward AND pointfeat's two other consumers (max over points, the rotation
        # heads' object-major copy) as one node: their three gradients meet in one pass (train_ops._PointfeatHub)
        g, pfmax, pf_obj = T.pointfeat_hub(pf, w("conv2.weight"), w("conv2.bias"), w("conv3.weight"), w("conv3.bias"),
                                           w("conv4.weight"), w("conv4.bias"), B, N, M,
                                           (buf["c2"], buf["c3"], buf["g"], buf["i"]), obj_copy=obj_copy)
        return g, pf, (pfmax, pf_obj)
    h = T.linear(pf, w("conv2.weight"), w("conv2.bias"), relu=True, pre=buf["c2"])
    h = T.linear(h, w("conv3.weight"), w("conv3.bias"), relu=True, pre=buf["c3"])
    g = T.linear_maxpool(h, w("conv4.weight"), w("conv4.bias"), False, B, N, M, pre=(buf["g"], buf["i"]))
    return g, pf, None


def fused_lp_ok(pts, p, N, M, prefix="pcl_net"):
    """The autocast fused encoder forward stores bf16 activation rows, which only the row-sparse chains read: every stack
    must take that form (it does at N, M multiples of 64 up to 4096 points per cloud, for inputs without gradient)."""
    w = lambda n: p[f"{prefix}.{n}"]
    h1_like = torch.empty(0, 64, device=pts.device)
    return (T.pooled_chain_ok(pts, w("stn.conv1.weight"), w("stn.conv2.weight"), w("stn.conv3.weight"), N, M)
            and T.pooled_chain_ok(h1_like, w("fstn.conv1.weight"), w("fstn.conv2.weight"), w("fstn.conv3.weight"), N, M)
            and T.pooled_chain_ok(h1_like, w("conv2.weight"), w("conv3.weight"), w("conv4.weight"), N, M))


def pointnet_rows(pts, p, B, N, M, feature_transform=True, prefix="pcl_net"):
    """PointNetfeat on cloud-major point rows [B*N + B*M, 3] -> (g [C,1024], pointfeat [R,64])  (pointnet.py:97-116)."""
    w = lambda n: p[f"{prefix}.{n}"]
    trans = _stn(pts, p, f"{prefix}.stn", 3, B, N, M)
    x1 = T.cloud_matmul(pts, trans, B, N, M, out_cols=8)                    # x^T @ trans, zero-padded to 8 columns
    if feature_transform:
        h1, h1s = T.linear_fan2(x1, w("conv1.weight"), w("conv1.bias"), relu=True)   # [R,64], two consumers
        trans_feat = _stn(h1s, p, f"{prefix}.fstn", 64, B, N, M)
        pf = T.cloud_matmul(h1, trans_feat, B, N, M)
    else:
        pf = T.linear(x1, w("conv1.weight"), w("conv1.bias"), relu=True)
    h = T.linear(pf, w("conv2.weight"), w("conv2.bias"), relu=True)
    h = T.linear(h, w("conv3.weight"), w("conv3.bias"), relu=True)
    g = T.linear_maxpool(h, w("conv4.weight"), w("conv4.bias"), False, B, N, M)   # conv4 has no ReLU (:114)
    return g, pf


def _rot_head(g, pf_obj, p, prefix, B, N, M, x_cm=False):
    """RotHead.forward (heads/conv_out_per_rot_head.py:126-140) on the never-materialised cat(pcl_feat,kps_feat).  x_cm: pf_obj
    is pointfeat in CLOUD-major row order (only the autocast one-node head takes that: forward_train decided so)."""
    w = lambda n: p[f"{prefix}.{n}"]
    P = N + M
    W0g, W0b = T.split_cols(w("layers.0.weight").reshape(256, 1088), 1024)   # global half (a view) | point half
    bias0 = T.linear(g, W0g, w("layers.0.bias"))                              # [2B,256]: global half + conv bias
    if T.rot_head_lp_ok(pf_obj, W0b, w("layers.3.weight"), w("layers.3.bias"), N, M):
        # autocast: the head as one node with bf16 [B*P,256] activations between its kernels
        from .heads import neck_weight3
        wn, bn = neck_weight3(w("neck.0.weight"), w("neck.0.bias"))
        return T.rot_head_lp(pf_obj, W0b, bias0, w("layers.1.weight"), w("layers.1.bias"), w("layers.3.weight"),
                             w("layers.3.bias"), w("layers.4.weight"), w("layers.4.bias"), wn, bn, w("conv_p.weight"),
                             p.get(f"{prefix}.conv_p.bias"), B, N, M, x_cm)[:, :w("neck.0.weight").shape[0]]
    assert not x_cm, "only the autocast one-node rotation head reads cloud-major rows"
    # [B*P,256]; the per-cloud bias and the GroupNorm tile partials are epilogue work of the GEMMs
    if T.rot_l0_block_ok(pf_obj, W0b, N, M):
        a = T.rot_l0_block(pf_obj, W0b, bias0, w("layers.1.weight"), w("layers.1.bias"), B, N, M)
    else:
        y, part = T.linear_cloudbias(pf_obj, W0b, bias0, B, N, M, with_gn_partials=True)
        a = T.gn_points_gelu(y, w("layers.1.weight"), w("layers.1.bias"), B, P, part)
    from .heads import neck_rows, neck_weight3

    rd = w("neck.0.weight").shape[0]
    if T.rot_l1_block_ok(a, w("layers.3.weight"), N, M) and w("layers.3.bias") is not None:
        wn, bn = neck_weight3(w("neck.0.weight"), w("neck.0.bias"))
        y3 = T.rot_l1_block(a, w("layers.3.weight"), w("layers.3.bias"), w("layers.4.weight"), w("layers.4.bias"), wn, bn,
                            B, N, M)
        return _first_cols(T.weighted_point_sum(y3, w("conv_p.weight"), p.get(f"{prefix}.conv_p.bias"), B, P), rd)
    if T.rot_l1_tail_lp_ok(a, w("layers.3.weight"), w("layers.3.bias"), N, M):
        # autocast: the whole second half of the head as one node whose backward is one pass on the bf16 matrix pipe
        wn, bn = neck_weight3(w("neck.0.weight"), w("neck.0.bias"))
        o = T.rot_l1_tail_lp(a, w("layers.3.weight"), w("layers.3.bias"), w("layers.4.weight"), w("layers.4.bias"), wn, bn,
                             w("conv_p.weight"), p.get(f"{prefix}.conv_p.bias"), B, N, M)
        return _first_cols(o, rd)
    y, part = T.linear_gn_partials(a, w("layers.3.weight"), w("layers.3.bias"), B, N, M)
    if part is not None and P % 64 == 0:
        # GroupNorm + GELU + neck + conv_p as one node: the [B*P,256] activation in between is never stored and the backward
        # needs no reduction pass over y (train_ops._NeckTail)
        wn, bn = neck_weight3(w("neck.0.weight"), w("neck.0.bias"))
        return _first_cols(T.neck_tail(y, w("layers.4.weight"), w("layers.4.bias"), wn, bn, w("conv_p.weight"),
                           p.get(f"{prefix}.conv_p.bias"), B, P, part), rd)
    else:
        a = T.gn_points_gelu(y, w("layers.4.weight"), w("layers.4.bias"), B, P, part)
        y3 = neck_rows(a, w("neck.0.weight"), w("neck.0.bias"))              # [B*P,3] (columns >= rot_dim are zero)
    return _first_cols(T.weighted_point_sum(y3, w("conv_p.weight"), p.get(f"{prefix}.conv_p.bias"), B, P), rd)


_ROT_PREFIX = ("rot_head.rot_head_x", "rot_head.rot_head_y")


def _first_cols(o, rd):
    """o[:, :rd] - the tensor itself when that is all of it (rot6d: rd = 3 of 3 columns; a slice node would cost a copy in
    the backward)."""
    return o if o.shape[1] == rd else o[:, :rd]


def _rot_heads_lp(g, pf, p, B, N, M, x_cm):
    """Both RotHeads under autocast as ONE node (train_ops._RotHeadPairLP): per head the kernels of `_RotHeadLP`, and the two
    data gradients pointfeat receives from them are summed by the second head's backward kernel instead of by autograd."""
    g = g if isinstance(g, (tuple, list)) else (g, g)   # one handle of the pooled feature per head (train_ops.hub)
    from .heads import neck_weight3

    heads = []
    for h, pre in enumerate(_ROT_PREFIX):
        w = lambda n: p[f"{pre}.{n}"]
        W0g, W0b = T.split_cols(w("layers.0.weight").reshape(256, 1088), 1024)
        bias0 = T.linear(g[h], W0g, w("layers.0.bias"))
        wn, bn = neck_weight3(w("neck.0.weight"), w("neck.0.bias"))
        heads.append((W0b, bias0, w("layers.1.weight"), w("layers.1.bias"), w("layers.3.weight"), w("layers.3.bias"),
                      w("layers.4.weight"), w("layers.4.bias"), wn, bn, w("conv_p.weight"), p.get(f"{pre}.conv_p.bias")))
    outs = T.rot_head_pair_lp(pf, heads[0], heads[1], B, N, M, x_cm)
    return [_first_cols(o, p[f"{pre}.neck.0.weight"].shape[0]) for pre, o in zip(_ROT_PREFIX, outs)]


def _rot_heads_shapes_ok_p(p, N, M):
    """:func:`_rot_heads_shapes_ok` before pointfeat exists (its width is pcl_net.conv1's: 64)."""
    return (tuple(p["pcl_net.conv1.weight"].shape[:1]) == (64,) and N % 64 == 0 and M % 64 == 0 and N > 0 and M > 0
            and all(p.get(f"{pre}.layers.3.bias") is not None
                    and tuple(p[f"{pre}.layers.0.weight"].shape[:2]) == (256, 1088)
                    and tuple(p[f"{pre}.layers.3.weight"].shape[:2]) == (256, 256) for pre in _ROT_PREFIX))


def _rot_heads_shapes_ok(p, pf_obj, N, M):
    return (pf_obj.shape[1] == 64 and N % 64 == 0 and M % 64 == 0 and N > 0 and M > 0
            and all(p.get(f"{pre}.layers.3.bias") is not None
                    and tuple(p[f"{pre}.layers.0.weight"].shape[:2]) == (256, 1088)
                    and tuple(p[f"{pre}.layers.3.weight"].shape[:2]) == (256, 256) for pre in _ROT_PREFIX))


def _rot_heads_fused_ok(p, pf_obj, N, M):
    return T.rot_heads_ok(pf_obj, N, M) and _rot_heads_shapes_ok(p, pf_obj, N, M)


def _rot_heads_split(g, pf, pf_obj, p, rt, B, N, M):
    """Both RotHeads in split mode: ONE fused forward launch chain (`catre_train_rot_fwd`, k_rot_l1_split<true>) computes what
    the per-head ops would - y0, a0 = gelu(GN0(y0)), y1 and the GroupNorm partials - and the per-head ops become graph nodes
    around those buffers (`pre=`); their backward is unchanged (split dgrad / wgrad GEMMs, fp32 GroupNorm / GELU passes)."""
    g = g if isinstance(g, (tuple, list)) else (g, g)   # one handle of the pooled feature per head (train_ops.hub)
    from .heads import neck_weight3

    P = N + M
    W0s, b0s = [], []
    for h, pre in enumerate(_ROT_PREFIX):
        W0g, W0l = T.split_cols(p[f"{pre}.layers.0.weight"].reshape(256, 1088), 1024)
        W0s.append(W0l)
        b0s.append(T.linear(g[h], W0g, p[f"{pre}.layers.0.bias"]))   # [2B,256]
    prm, packed = rt._train_packs(pf.device, 2)
    buf = T.rot_heads_forward(pf.detach(), b0s[0], b0s[1], prm, packed, B, N, M, 2)
    out = []
    for h, pre in enumerate(_ROT_PREFIX):
        w = lambda n: p[f"{pre}.{n}"]
        if T.knobs().split_l0_one_pass:
            # the first block as one node: its backward is the fp32 one-pass kernel (k_rot_l0_bwd: sums + one pass over
            # (da, y0)) instead of the GroupNorm apply pass, a per-cloud bias reduction and a split dgrad and wgrad
            a = T.rot_l0_block(pf_obj, W0s[h], b0s[h], w("layers.1.weight"), w("layers.1.bias"), B, N, M,
                               pre=(buf["y0"][h], buf["a0"][h], buf["stat0"][h]))
        else:
            y, _ = T.linear_cloudbias(pf_obj, W0s[h], b0s[h], B, N, M, with_gn_partials=True,
                                      pre=(buf["y0"][h], None))
            a = T.gn_points_gelu(y, w("layers.1.weight"), w("layers.1.bias"), B, P, None,
                                 pre=(buf["a0"][h], buf["stat0"][h]))
        wn, bn = neck_weight3(w("neck.0.weight"), w("neck.0.bias"))
        rd = w("neck.0.weight").shape[0]
        if T.knobs().split_l1_one_pass:
            # the second block + tail as one node: backward = conv_p, then ONE pass with hi + lo operands (k_rot_l1_bwd_sp)
            o = T.rot_l1_tail_lp(a, w("layers.3.weight"), w("layers.3.bias"), w("layers.4.weight"), w("layers.4.bias"),
                                 wn, bn, w("conv_p.weight"), p.get(f"{pre}.conv_p.bias"), B, N, M,
                                 pre=(buf["y1"][h], buf["part1"][h]))
            out.append(_first_cols(o, rd))
            continue
        y1, part1 = T.linear_gn_partials(a, w("layers.3.weight"), w("layers.3.bias"), B, N, M,
                                         pre=(buf["y1"][h], buf["part1"][h]))
        out.append(_first_cols(T.neck_tail(y1, w("layers.4.weight"), w("layers.4.bias"), wn, bn, w("conv_p.weight"),
                               p.get(f"{pre}.conv_p.bias"), B, P, part1), rd))
    return out


def _rot_heads_fused(g, pf, pf_obj, p, rt, B, N, M):
    """Both RotHeads (heads/conv_out_per_rot_head.py:126-140) with the fused forward (train_ops._RotHeads, fp32)."""
    g = g if isinstance(g, (tuple, list)) else (g, g)   # one handle of the pooled feature per head (train_ops.hub)
    from .heads import neck_weight3

    heads = []
    for h, pre in enumerate(_ROT_PREFIX):
        w = lambda n: p[f"{pre}.{n}"]
        W0g, W0l = T.split_cols(w("layers.0.weight").reshape(256, 1088), 1024)   # global half | point half
        bias0 = T.linear(g[h], W0g, w("layers.0.bias"))                           # [2B,256]: global half + conv bias
        wn, bn = neck_weight3(w("neck.0.weight"), w("neck.0.bias"))
        heads.append((bias0, W0l, w("layers.1.weight"), w("layers.1.bias"), w("layers.3.weight"),
                      w("layers.3.bias"), w("layers.4.weight"), w("layers.4.bias"), wn, bn, w("conv_p.weight"),
                      p.get(f"{pre}.conv_p.bias")))
    prm, packed = rt._train_packs(pf.device, 0)
    outs = T.rot_heads(pf.detach(), pf_obj, prm, packed, B, N, M, heads[0], heads[1])
    return [_first_cols(o, p[f"{pre}.neck.0.weight"].shape[0]) for pre, o in zip(_ROT_PREFIX, outs)]


def forward_train(p, opts, x, tfd_kps, init_pose, init_scale, K_zoom=None, mean_scales=None, rt=None):
    """p: {state_dict key: live parameter}.  Returns (pose [B,3,4], scale [B,3], aux dict) - autograd-connected.
    ``rt``: the model's :class:`~catre_amd.runtime.HipRuntime`; with it the fp32 encoder forward takes the fused kernels."""
    B, N, M = x.shape[0], x.shape[2], tfd_kps.shape[2]
    hip.require_dev_f32(x, "x", (B, 3, N), contiguous=False)
    hip.require_dev_f32(tfd_kps, "tfd_kps", (B, 3, M), contiguous=False)
    # one decision, taken once: the fused fp32 rotation heads read pointfeat cloud-major (forward AND backward), so no
    # object-major copy is made for them
    fused_rot = (rt is not None and T._amp() == 0 and bool(opts.feature_transform) and N + M == rt.N + rt.M
                 and _rot_heads_shapes_ok_p(p, N, M))
    # autocast: each head is one node (train_ops._RotHeadLP) that reads pointfeat cloud-major as well - no object-major copy
    lp_cm = (rt is not None and T._amp() == 1 and T.knobs().fused_lp_rot and T.knobs().lp_rot_bf16_rows
             and bool(opts.feature_transform) and N + M == rt.N + rt.M and _rot_heads_shapes_ok_p(p, N, M))
    enc_frozen = (not x.requires_grad and not tfd_kps.requires_grad
                  and not any(v.requires_grad for k, v in p.items() if k.startswith("pcl_net.")))
    if rt is not None and enc_frozen and T._amp() == 0 and N + M == rt.N + rt.M:
        # PCLNET.FREEZE (CATRE_disR_shared.py:301-304): nothing in front of the heads needs a gradient, so the encoder runs
        # on the INFERENCE kernels - no activation saves, no arg-max rows, no graph nodes (and no encoder backward)
        # like train_stn3d: the first kernel of a training forward re-packs what it reads - the fingerprint cannot see writes
        # through `p.data` (the reference's own Ranger, EMA), and the heads' backward reads the LIVE w0 / w1
        rt._fingerprint = None
        st = rt.stage_pointnet(x, tfd_kps, feature_transform=bool(opts.feature_transform))
        g, pfmax, pf = st["gfeat"][:, :1024], st["gfeat"][:, 1024:], st["pointfeat"]
        hub = (pfmax, pf if fused_rot else T.object_major(pf, B, N, M))
    elif rt is not None and T._amp() in (0, 1, 2) and opts.feature_transform and N % 64 == 0 and M % 64 == 0 \
            and N + M == rt.N + rt.M:
        pts = _cloud_major_rows(x, tfd_kps)
        if T._amp() == 1 and not fused_lp_ok(pts, p, N, M):
            (g, pf), hub = pointnet_rows(pts, p, B, N, M, True), None   # (e.g. a differentiable 3-d input: layer-wise ops)
        else:
            g, pf, hub = pointnet_rows_fused(pts, hip.points_desc(x, tfd_kps), rt, p, B, N, M, mode=T._amp(),
                                             obj_copy=not (fused_rot or lp_cm))
        fused_rot = fused_rot and hub is not None
        lp_cm = lp_cm and hub is not None
    else:
        pts = _cloud_major_rows(x, tfd_kps)
        (g, pf), hub = pointnet_rows(pts, p, B, N, M, bool(opts.feature_transform)), None
        fused_rot = lp_cm = False
    # max_n pointfeat (flat_pcl_feat tail) and the rot heads' input in object-major order: [N observed | M prior] per object
    # (CATRE_disR_shared.py:69, :86)
    pfmax, pf_obj = hub if hub is not None else (T.maxpool_points(pf, B, N, M), T.object_major(pf, B, N, M))

    # g (pooled feature, [2B,1024]) has three consumers - its first B rows go to the ts head, all of it to each rotation
    # head - and pfmax one that reads its first B rows: their gradients are summed by one launch each (train_ops._Hub)
    if opts.with_kps_feature or g.shape[0] == B:
        feats = [g[:B], pfmax[:B]] + ([g[B:], pfmax[B:]] if opts.with_kps_feature else [])
        gs = (g, g)
    else:
        g_ts, gx, gy = T.hub(g, B)
        feats = [g_ts, T.hub(pfmax, B)[0]]
        gs = (gx, gy)
    if opts.with_init_scale:
        feats.append(init_scale)
    if opts.with_init_trans:
        feats.append(init_pose[:, :3, 3])
    ts_feat = torch.cat(feats, 1)                                             # [B, ts_in]
    h = T.linear(ts_feat, p["ts_head.linears.0.weight"], p["ts_head.linears.0.bias"])
    h = T.gn_rows_gelu(h, p["ts_head.linears.1.weight"], p["ts_head.linears.1.bias"])
    h = T.linear(h, p["ts_head.linears.3.weight"], p["ts_head.linears.3.bias"])
    h = T.gn_rows_gelu(h, p["ts_head.linears.4.weight"], p["ts_head.linears.4.bias"])
    dt = T.linear(h, p["ts_head.fc_t.weight"], p["ts_head.fc_t.bias"])
    ds = T.linear(h, p["ts_head.fc_s.weight"], p["ts_head.fc_s.bias"])

    if fused_rot:
        rx, ry = _rot_heads_fused(gs, pf, pf_obj, p, rt, B, N, M)
    elif hub is not None and T._amp() == 2 and _rot_heads_shapes_ok(p, pf_obj, N, M):
        rx, ry = _rot_heads_split(gs, pf, pf_obj, p, rt, B, N, M)
    elif lp_cm:
        rx, ry = _rot_heads_lp(gs, pf_obj, p, B, N, M, True)
    else:
        rx = _rot_head(gs[0], pf_obj, p, "rot_head.rot_head_x", B, N, M, lp_cm)
        ry = _rot_head(gs[1], pf_obj, p, "rot_head.rot_head_y", B, N, M, lp_cm)
    rot6d = torch.cat([rx, ry], 1)

    pose, scale = T.pose_update_autograd(rot6d, dt, ds, init_pose, init_scale, mean_scales, K_zoom, opts)
    return pose, scale, dict(rot6d=rot6d, trans_deltas=dt, scale_deltas=ds)
