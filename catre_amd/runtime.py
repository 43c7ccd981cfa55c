"""Host-side runtime: binds a set of parameter tensors to the C ABI.

* keeps the ``const float* const* params`` array alive and in ``catre_param`` order,
* re-packs the MFMA weight image (``catre_pack_weights``) only when a parameter changed
  (``tensor._version`` / ``data_ptr`` fingerprint), stream-ordered with the forward that needs it,
* owns a grow-only workspace per device (sized by ``catre_workspace_bytes``; 288 GB of HBM3E per
  MI355X means B=256, N=M=1024 needs ~1.4 GB and is simply kept resident).

No torch op is on the hot path: the only torch calls are ``torch.empty`` for outputs/workspace.
"""
import ctypes
import logging

import torch

from . import hip

logger = logging.getLogger(__name__)


def opts_from_cfg(cfg, feature_transform=True):
    """Translate the cfg flags read by ``CATRE_disR_shared.forward`` (reference
    ``core/catre/models/CATRE_disR_shared.py:57-120``) into ``catre_opts``."""
    net = cfg.MODEL.CATRE
    rh, th = net.ROT_HEAD, net.TS_HEAD
    rot_type = hip.rot_type_id(rh.ROT_TYPE)  # ValueError on an unknown name, like get_rot_dim (model_utils.py:24)
    if rh.get("CLASS_AWARE", False):
        # the reference branch itself is unreachable: it dereferences the non-existent self.pose_head
        # (CATRE_disR_shared.py:90-95)
        raise NotImplementedError("ROT_HEAD.CLASS_AWARE=True is not supported (latent bug in the reference too)")
    if rh.DELTA_T_SPACE not in ("image", "3D"):
        raise ValueError("Unknown delta_T_space: {}".format(rh.DELTA_T_SPACE))  # pose_scale_from_delta_init.py:76
    o = hip.CatreOpts()
    o.feature_transform = int(bool(feature_transform))
    o.with_kps_feature = int(bool(th.WITH_KPS_FEATURE))
    o.with_init_scale = int(bool(th.WITH_INIT_SCALE))
    o.with_init_trans = int(bool(th.get("WITH_INIT_TRANS", False)))
    o.delta_t_space_3d = int(rh.DELTA_T_SPACE == "3D")
    o.delta_z_deepim = int(rh.DELTA_Z_STYLE != "cosypose")
    o.k_aware = int(bool(rh.T_TRANSFORM_K_AWARE))
    o.scale_mul = int("add" not in rh.SCLAE_TYPE)
    o.scale_base_mean = int("iter" not in rh.SCLAE_TYPE)
    o.is_allo = int("allo" in rh.ROT_TYPE)
    o.refine_scale = int(bool(cfg.MODEL.REFINE_SCLAE))
    o.zero_center = int(bool(cfg.INPUT.ZERO_CENTER_INPUT))
    o.delta_t_weight = float(rh.DELTA_T_WEIGHT)
    o.allo_eps = 1e-4  # CATRE_disR_shared.py:112
    o.ts_in_dim = 1088 * (2 if o.with_kps_feature else 1) + 3 * o.with_init_scale + 3 * o.with_init_trans
    o.rot_type = rot_type
    return o


class HipRuntime:
    """One per model instance (and device).

    One runtime serves ONE stream at a time for weight packing: the packed image is a single buffer that ``params``
    rewrites in place, stream-ordered, whenever a parameter changed - and the training forward rewrites it on every
    call (``train_stn3d``).  Inference calls on several streams may share a runtime (each stream has its own workspace,
    the image is only read); a training stream next to a concurrent inference stream - or a ``GraphedRefine`` replay -
    on the SAME runtime is not supported: use a second model instance (``copy.deepcopy`` drops the runtime)."""

    def __init__(self, named_params, N, M, ts_in_dim, root=None):
        """``named_params``: callable returning ``{state_dict key: tensor}`` of the LIVE parameters.  ``root`` (optional):
        the module whose ``named_parameters()`` keys are the state_dict keys as they are - lets ``_live_params`` read the
        parameters through cached ``module._parameters`` slots instead of walking the module tree on every call (the walk
        was 60 % of a forward's host time at B=1, profiles/eval_loop_cprofile.py)."""
        self._named_params = named_params
        self._root = root
        self._slots = None
        self._links = None
        self.N, self.M, self.ts_in_dim = int(N), int(M), int(ts_in_dim)
        self._fingerprint = None
        self._param_arr = None
        self._param_keep = None
        self._packed = None
        self._packed_sel = 0
        self._capture_fp = None  # ((stream, capture id, fingerprint), sel) of the pack recorded by the capture in progress
        self._ws = None

    # ------------------------------------------------------------------ weights
    def _live_params(self):
        if self._root is None:
            named = self._named_params()
            return [named.get(k) for k in hip.PARAM_KEYS]
        # The live Parameter objects, read through the owning modules' `_parameters` dicts: a re-assigned parameter
        # (`m.weight = nn.Parameter(...)`) is seen because the lookup goes through the dict, a replaced sub-module
        # because every parent -> child link of the tree is re-checked (identity) first.
        links = self._links
        if links is None or not all(d.get(n) is c for d, n, c in links):
            root = self._root
            self._links = [(parent._modules, n, c) for parent in root.modules() for n, c in parent._modules.items()
                           if c is not None]
            slot = {}
            for prefix, mod in root.named_modules():
                for pn in mod._parameters:
                    slot[f"{prefix}.{pn}" if prefix else pn] = (mod._parameters, pn)
            self._slots = [slot.get(k) for k in hip.PARAM_KEYS]
        return [s[0].get(s[1]) if s is not None else None for s in self._slots]

    def params(self, device, sel=hip.PACK_ALL):
        """(param pointer array, packed weights) - re-packed on the current stream if stale.  ``sel``: the packs the
        caller needs (``hip.PACK_*``); a training step, which re-packs after every optimizer step, asks for the fp32
        encoder image only."""
        lib = hip.load()
        tensors = self._live_params()
        fp = (hip.param_epoch(),) + tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors)
        capturing = torch.cuda.is_current_stream_capturing()
        # keyed on the capture's identity, not only on (stream, weights): a second capture on the same stream with unchanged
        # weights is another graph and needs its own pack node
        cap_key = (torch.cuda.current_stream(device).cuda_stream, hip.capture_id(device), fp) if capturing else None
        if capturing and self._packed is not None and self._packed.device == device and self._capture_fp is not None \
                and self._capture_fp[0] == cap_key and not (sel & ~self._capture_fp[1]):
            # this capture has already recorded a pack of these weights holding every image asked for: the later
            # encoder / head entry points of the same captured forward reuse it (one pack node per replay, not one per call)
            return self._param_arr, self._packed
        if not capturing:
            self._capture_fp = None
        if fp == self._fingerprint and self._packed is not None and self._packed.device == device \
                and (sel & ~self._packed_sel):
            sel |= self._packed_sel  # same weights, more packs wanted: redo with the union
            self._fingerprint = None
        if fp != self._fingerprint or self._packed is None or self._packed.device != device:
            for k, t in zip(hip.PARAM_KEYS, tensors):
                if t is not None and t.device != device:
                    raise hip.CatreHipError(f"parameter {k} is on {t.device}, inputs are on {device}")
            tensors = [t.detach().contiguous() if t is not None else None for t in tensors]
            self._param_keep = tensors
            self._param_arr = hip.param_array(tensors)
            n = lib.catre_packed_floats(self.N, self.M, self.ts_in_dim)
            if self._packed is None or self._packed.numel() < n or self._packed.device != device:
                self._packed = torch.empty(n, dtype=torch.float32, device=device)
            hip.check(
                lib.catre_pack_weights_sel(self._param_arr, self.N, self.M, self.ts_in_dim, hip.ptr(self._packed),
                                           self._packed.numel(), int(sel), hip.stream_ptr(device)),
                "catre_pack_weights_sel",
            )
            # a pack issued while the stream is being captured is only RECORDED (GraphedTrainStep): the buffer still holds
            # whatever the last executed pack wrote, so the cache must not call it fresh
            self._fingerprint = None if capturing else fp
            self._packed_sel = int(sel)
            if capturing:
                self._capture_fp = (cap_key, int(sel))
        return self._param_arr, self._packed

    # ------------------------------------------------------------------ workspace
    def workspace(self, B, N, M, device):
        lib = hip.load()
        need = lib.catre_workspace_bytes(B, N, M)
        if need == 0:
            raise ValueError(f"bad sizes B={B} N={N} M={M}")
        # one scratch buffer per (device, stream): calls issued on different streams (several images refined
        # concurrently, profiles/multi_stream_probe.py) must not share intermediates; calls on one stream are ordered
        if self._ws is None:
            self._ws = {}
        key = (device.index, hip.stream_ptr(device).value or 0)
        ws = self._ws.pop(key, None)
        if ws is None or ws.numel() < need:
            if ws is None and len(self._ws) >= 16:
                # least recently used stream only (dicts keep insertion order; every use re-inserts): the caching
                # allocator keeps the block alive until the work queued on it is done, and whoever captured an address
                # (GraphedRefine / GraphedTrainStep) holds its own reference
                old = next(iter(self._ws))
                del self._ws[old]
                logger.info("HipRuntime: workspace of stream %#x evicted (16 streams cached)", old[1])
            ws = torch.empty(need, dtype=torch.uint8, device=device)
        self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ drivers
    def refine_iter(self, x, tfd_kps, init_pose, init_scale, K_zoom, mean_scales, opts):
        """One ``CATRE_disR_shared.forward`` (test path) -> (pose [B,3,4], scale [B,3])."""
        lib = hip.load()
        pts = hip.points_desc(x, tfd_kps)
        B, N, M = x.shape[0], x.shape[2], tfd_kps.shape[2]
        dev = x.device
        self._check_nm(N, M)
        init_pose = hip.require_dev_f32(init_pose.contiguous(), "init_pose", (B, 3, 4))
        init_scale = hip.require_dev_f32(init_scale.contiguous(), "init_scale", (B, 3))
        Ks = hip.require_dev_f32(K_zoom.contiguous(), "K_zoom", (B, 3, 3)) if K_zoom is not None else None
        ms = hip.require_dev_f32(mean_scales.contiguous(), "mean_scales", (B, 3)) if mean_scales is not None else None
        if opts.k_aware and not opts.delta_t_space_3d:
            assert Ks is not None and Ks.shape == (B, 3, 3)  # pose_scale_from_delta_init.py:64
        if opts.scale_base_mean and ms is None:
            raise ValueError("SCLAE_TYPE without 'iter' needs mean_scales")
        prm, packed = self.params(dev)
        ws = self.workspace(B, N, M, dev)
        pose_out = torch.empty(B, 3, 4, dtype=torch.float32, device=dev)
        scale_out = torch.empty(B, 3, dtype=torch.float32, device=dev)
        hip.check(
            lib.catre_refine_iter(ctypes.byref(pts), hip.ptr(init_pose), hip.ptr(init_scale), hip.ptr(ms), hip.ptr(Ks),
                                  prm, hip.ptr(packed), ctypes.byref(opts), hip.ptr(pose_out), hip.ptr(scale_out),
                                  hip.ptr(ws), ws.numel(), B, N, M, hip.stream_ptr(dev)),
            "catre_refine_iter",
        )
        return pose_out, scale_out

    def refine_k(self, pcl, obj_kps, init_pose, init_scale, K, mean_scales, opts, n_iter):
        """Fused K-loop -> poses [n_iter+1,B,3,4], scales [n_iter+1,B,3] (slot 0 = initial estimate)."""
        lib = hip.load()
        B, N, M = pcl.shape[0], pcl.shape[1], obj_kps.shape[1]
        dev = pcl.device
        self._check_nm(N, M)
        pcl = hip.require_dev_f32(pcl.contiguous(), "pcl", (B, N, 3))
        obj_kps = hip.require_dev_f32(obj_kps.contiguous(), "obj_kps", (B, M, 3))
        hip.require_dev_f32(init_pose, "init_pose", (B, 3, 4), contiguous=False)
        hip.require_dev_f32(init_scale, "init_scale", (B, 3), contiguous=False)
        Ks = hip.require_dev_f32(K.contiguous(), "K", (B, 3, 3)) if K is not None else None
        ms = hip.require_dev_f32(mean_scales.contiguous(), "mean_scales", (B, 3)) if mean_scales is not None else None
        if opts.k_aware and not opts.delta_t_space_3d and Ks is None:
            raise ValueError("T_TRANSFORM_K_AWARE needs K")
        if opts.scale_base_mean and ms is None:
            raise ValueError("SCLAE_TYPE without 'iter' needs mean_scales")
        prm, packed = self.params(dev)
        ws = self.workspace(B, N, M, dev)
        poses = torch.empty(n_iter + 1, B, 3, 4, dtype=torch.float32, device=dev)
        scales = torch.empty(n_iter + 1, B, 3, dtype=torch.float32, device=dev)
        # slot 0 (the reference's out_dict["pose_0"], catre_evaluator.py:292) is written by iteration 1's pose-update kernel:
        # no copy launch inside a refine
        init_pose, init_scale = init_pose.contiguous(), init_scale.contiguous()
        hip.check(
            lib.catre_refine_k_from(hip.ptr(pcl), hip.ptr(obj_kps), hip.ptr(init_pose), hip.ptr(init_scale), hip.ptr(ms),
                                    hip.ptr(Ks), prm, hip.ptr(packed), ctypes.byref(opts), hip.ptr(poses), hip.ptr(scales),
                                    hip.ptr(ws), ws.numel(), B, N, M, n_iter, hip.stream_ptr(dev)),
            "catre_refine_k_from",
        )
        return poses, scales

    def _check_nm(self, N, M):
        if N + M != self.N + self.M:
            # conv_p bakes the number of points into the weights (conv_out_per_rot_head.py:112)
            raise ValueError(
                f"got N+M={N + M} points but the rotation head was built for num_points={self.N + self.M}"
            )

    # ------------------------------------------------------------------ training forward on the fused encoder kernels
    def train_encoder_buffers(self, B, N, M, device, stn_rows=True, mode=0):
        """stn_rows=False (fp32): the STN stacks' activation rows are not stored - their backward recomputes them on its
        live rows (train_ops._PooledChain, catre_op_stn_recompute): 0.94 GB less at B = 256.
        mode 1 (the bf16-operand kernels): the rows of the three conv stacks behind their first layer (a1 / a2, f1 / f2,
        c2 / c3) are bf16 - those kernels hold them as bf16 and the fp32 rows they used to write held the same values."""
        R, C = B * (N + M), 2 * B
        e = lambda *shape, dt=torch.float32: torch.empty(*shape, dtype=dt, device=device)
        rdt = torch.bfloat16 if int(mode) == 1 else torch.float32
        buf = dict(
            g_stn=e(C, 1024), i_stn=e(C, 1024, dt=torch.int32), g_fstn=e(C, 1024), i_fstn=e(C, 1024, dt=torch.int32),
            x1=e(R, 8), h1=e(R, 64), pf=e(R, 64), c2=e(R, 128, dt=rdt), c3=e(R, 512, dt=rdt), g=e(C, 1024),
            i=e(C, 1024, dt=torch.int32))
        for k, w in (("a1", 64), ("a2", 128), ("f1", 64), ("f2", 128)):
            buf[k] = e(R, w, dt=rdt) if stn_rows else None
        return buf

    def _train_packs(self, device, mode):
        """(params, packed) for the fused training forward in `mode` (0 = fp32 kernels: encoder AND head images, the rotation
        heads' fused forward reads the latter; 1 = bf16-operand kernels; 2 = split: its trunk keeps conv2 as an fp32 MFMA
        layer, so it reads the fp32 encoder image as well)."""
        return self.params(device, {0: hip.PACK_F32_ENCODER | hip.PACK_F32_HEADS, 1: hip.PACK_BF16,
                                    2: hip.PACK_SPLIT | hip.PACK_F32_ENCODER}[int(mode)])

    def train_stn3d(self, pts, buf, B, N, M, device, mode=0):
        lib = hip.load()
        # first encoder kernel of a training forward: always re-pack the encoder image it reads (one ~6 us launch).  The
        # (data_ptr, _version, epoch) fingerprint cannot see writes through `p.data` (EMA, third-party optimizers), and a
        # stale forward image next to a live-weight backward would give inconsistent gradients without any error.
        self._fingerprint = None
        prm, packed = packs = self._train_packs(device, mode)
        ws = self.workspace(B, N, M, device)
        hip.check(lib.catre_train_stn3d_fwd(ctypes.byref(pts), prm, hip.ptr(packed), hip.ptr(buf["a1"]), hip.ptr(buf["a2"]),
                                            hip.ptr(buf["g_stn"]), hip.ptr(buf["i_stn"]), hip.ptr(ws), ws.numel(), B, N, M,
                                            int(mode), hip.stream_ptr(device)), "catre_train_stn3d_fwd")
        return packs   # what the other two encoder kernels of the SAME forward may be handed (`packs=`: no second fingerprint)

    def train_stnkd(self, pts, trans3, buf, B, N, M, device, mode=0, packs=None):
        lib = hip.load()
        prm, packed = packs if packs is not None else self._train_packs(device, mode)
        ws = self.workspace(B, N, M, device)
        hip.check(lib.catre_train_stnkd_fwd(ctypes.byref(pts), hip.ptr(trans3), prm, hip.ptr(packed), hip.ptr(buf["f1"]),
                                            hip.ptr(buf["f2"]), hip.ptr(buf["g_fstn"]), hip.ptr(buf["i_fstn"]), hip.ptr(ws),
                                            ws.numel(), B, N, M, int(mode), hip.stream_ptr(device)), "catre_train_stnkd_fwd")

    def train_trunk(self, pts, trans3, trans64, buf, B, N, M, device, mode=0, packs=None):
        lib = hip.load()
        prm, packed = packs if packs is not None else self._train_packs(device, mode)
        ws = self.workspace(B, N, M, device)
        hip.check(lib.catre_train_trunk_fwd(ctypes.byref(pts), hip.ptr(trans3), hip.ptr(trans64), prm, hip.ptr(packed),
                                            hip.ptr(buf["x1"]), hip.ptr(buf["h1"]), hip.ptr(buf["pf"]), hip.ptr(buf["c2"]),
                                            hip.ptr(buf["c3"]), hip.ptr(buf["g"]), hip.ptr(buf["i"]), hip.ptr(ws),
                                            ws.numel(), B, N, M, int(mode), hip.stream_ptr(device)), "catre_train_trunk_fwd")

    # ------------------------------------------------------------------ single stages (tests, sub-modules)
    def stage_linear(self, x, W, bias, relu=False, add_identity_k=0):
        """y = act(x W^T + b) (+ I_k) through ``catre_linear`` (F.linear as used by pointnet.py:31-40)."""
        lib = hip.load()
        R, K = x.shape
        J = W.shape[0]
        y = torch.empty(R, J, dtype=torch.float32, device=x.device)
        hip.check(
            lib.catre_linear(hip.ptr(x), x.stride(0), hip.ptr(W), W.stride(0), hip.ptr(bias), hip.ptr(y), J, R, J, K,
                             int(relu), int(add_identity_k), hip.stream_ptr(x.device)),
            "catre_linear",
        )
        return y

    def _stn_tail(self, pooled, named, prefix, k):
        h = self.stage_linear(pooled, named[f"{prefix}.fc1.weight"], named[f"{prefix}.fc1.bias"], relu=True)
        h = self.stage_linear(h, named[f"{prefix}.fc2.weight"], named[f"{prefix}.fc2.bias"], relu=True)
        return self.stage_linear(h, named[f"{prefix}.fc3.weight"], named[f"{prefix}.fc3.bias"], add_identity_k=k)

    def stage_pointnet(self, x, tfd_kps=None, feature_transform=True):
        """PointNetfeat stages on x [B,3,N] (and optionally tfd_kps [B,3,M]) -> dict with
        ``trans [C,3,3]``, ``trans_feat [C,64,64]``, ``stn_pool``/``fstn_pool [C,1024]``,
        ``gfeat [C,1088]``, ``pointfeat`` (flat [B*(N+M),64], observed clouds first); C = B or 2B."""
        lib = hip.load()
        dev = x.device
        B, N = x.shape[0], x.shape[2]
        if tfd_kps is None:
            M = 0
            pts = hip.points_desc(x, x)
        else:
            M = tfd_kps.shape[2]
            pts = hip.points_desc(x, tfd_kps)
        C = 2 * B if M > 0 else B
        prm, packed = self.params(dev)
        named = {k: t for k, t in zip(hip.PARAM_KEYS, self._param_keep)}
        ws = self.workspace(B, N, M, dev)
        st = hip.stream_ptr(dev)
        out = {}
        pool = torch.empty(C, 1024, dtype=torch.float32, device=dev)
        hip.check(lib.catre_stn3d_pool(ctypes.byref(pts), prm, hip.ptr(packed), hip.ptr(pool), hip.ptr(ws), ws.numel(),
                                       B, N, M, st), "catre_stn3d_pool")
        out["stn_pool"] = pool
        trans = self._stn_tail(pool, named, "pcl_net.stn", 3)
        out["trans"] = trans.view(C, 3, 3)
        t64 = None
        if feature_transform:
            pool2 = torch.empty(C, 1024, dtype=torch.float32, device=dev)
            hip.check(lib.catre_stnkd_pool(ctypes.byref(pts), hip.ptr(trans), prm, hip.ptr(packed), hip.ptr(pool2),
                                           hip.ptr(ws), ws.numel(), B, N, M, st), "catre_stnkd_pool")
            out["fstn_pool"] = pool2
            t64 = self._stn_tail(pool2, named, "pcl_net.fstn", 64)
            out["trans_feat"] = t64.view(C, 64, 64)
        gfeat = torch.empty(C, 1088, dtype=torch.float32, device=dev)
        pointfeat = torch.empty(B * (N + M), 64, dtype=torch.float32, device=dev)
        hip.check(lib.catre_trunk(ctypes.byref(pts), hip.ptr(trans), hip.ptr(t64), prm, hip.ptr(packed), hip.ptr(gfeat),
                                  hip.ptr(pointfeat), hip.ptr(ws), ws.numel(), B, N, M, st), "catre_trunk")
        out["gfeat"], out["pointfeat"] = gfeat, pointfeat
        return out

    def stage_ts_head(self, gfeat, init_pose, init_scale, opts):
        lib = hip.load()
        dev = gfeat.device
        B = init_scale.shape[0]
        prm, packed = self.params(dev)
        dt = torch.empty(B, 3, dtype=torch.float32, device=dev)
        ds = torch.empty(B, 3, dtype=torch.float32, device=dev)
        init_pose, init_scale = init_pose.contiguous(), init_scale.contiguous()  # bound: must outlive the enqueue
        ws = self.workspace(B, self.N, self.M, dev)
        hip.check(lib.catre_ts_head(hip.ptr(gfeat), hip.ptr(init_pose), hip.ptr(init_scale),
                                    prm, hip.ptr(packed), ctypes.byref(opts), hip.ptr(dt), hip.ptr(ds), hip.ptr(ws),
                                    ws.numel(), B, hip.stream_ptr(dev)), "catre_ts_head")
        return dt, ds

    def stage_rot_head(self, gfeat, pointfeat, B, N, M, rot_dim=3):
        lib = hip.load()
        dev = gfeat.device
        prm, packed = self.params(dev)
        ws = self.workspace(B, N, M, dev)
        rot6d = torch.empty(B, 2 * rot_dim, dtype=torch.float32, device=dev)
        hip.check(lib.catre_rot_head_dim(hip.ptr(gfeat), hip.ptr(pointfeat), prm, hip.ptr(packed), hip.ptr(rot6d),
                                         hip.ptr(ws), ws.numel(), B, N, M, int(rot_dim), hip.stream_ptr(dev)),
                  "catre_rot_head_dim")
        return rot6d


def pose_apply(pcl, obj_kps, pose, scale, zero_center=True):
    """``batch_updater_test`` core (engine/batch_test.py:81-97) -> x [B,3,N], tfd_kps [B,3,M] as permuted
    views of point-major buffers, exactly the layout the reference hands to the model."""
    lib = hip.load()
    B, N, M = pcl.shape[0], pcl.shape[1], obj_kps.shape[1]
    dev = pcl.device
    pcl = hip.require_dev_f32(pcl.contiguous(), "pcl", (B, N, 3))
    obj_kps = hip.require_dev_f32(obj_kps.contiguous(), "obj_kps", (B, M, 3))
    pose = hip.require_dev_f32(pose.contiguous(), "pose", (B, 3, 4))
    scale = hip.require_dev_f32(scale.contiguous(), "scale", (B, 3))
    # one buffer, observed rows then prior rows: the training forward reads the two as ONE cloud-major [B*N + B*M, 3] matrix
    # (train_forward._cloud_major_rows) without a cat
    both = torch.empty(B * (N + M), 3, dtype=torch.float32, device=dev)
    xo, ko = both[: B * N].view(B, N, 3), both[B * N:].view(B, M, 3)
    hip.check(lib.catre_pose_apply(hip.ptr(pcl), hip.ptr(obj_kps), hip.ptr(pose), hip.ptr(scale), hip.ptr(xo),
                                   hip.ptr(ko), B, N, M, int(zero_center), hip.stream_ptr(dev)), "catre_pose_apply")
    return xo.permute(0, 2, 1), ko.permute(0, 2, 1)


def pose_update(rot6d, trans_deltas, scale_deltas, init_pose, init_scale, mean_scales, Ks, opts):
    lib = hip.load()
    B = rot6d.shape[0]
    dev = rot6d.device
    pose_out = torch.empty(B, 3, 4, dtype=torch.float32, device=dev)
    scale_out = torch.empty(B, 3, dtype=torch.float32, device=dev)
    # contiguous copies stay bound until the launch is enqueued (a temporary's block would be recycled by the next copy)
    keep = [t.contiguous() if t is not None else None
            for t in (rot6d, trans_deltas, scale_deltas, init_pose, init_scale, mean_scales, Ks)]
    hip.check(lib.catre_pose_update(*[hip.ptr(t) for t in keep], ctypes.byref(opts), hip.ptr(pose_out), hip.ptr(scale_out), B,
                                    hip.stream_ptr(dev)), "catre_pose_update")
    del keep
    return pose_out, scale_out


def colmax(x):
    """``torch.max(x, 2)[0]`` for x [B,C,N] on the HIP device (stand-alone max-pool kernel)."""
    lib = hip.load()
    x = hip.require_dev_f32(x, "x", (None, None, None))
    B, C, N = x.shape
    out = torch.empty(B, C, dtype=torch.float32, device=x.device)
    hip.check(lib.catre_colmax(hip.ptr(x), hip.ptr(out), B, C, N, hip.stream_ptr(x.device)), "catre_colmax")
    return out


def rot_to_mat(rot, rot_type_id):
    """``get_rot_mat`` on the device (reference models/model_utils.py:28-40): rot [B,d] -> R [B,3,3]."""
    lib = hip.load()
    d = hip.ROT_DIMS[rot_type_id]
    rot = hip.require_dev_f32(rot.contiguous(), "rot", (None, d))
    B = rot.shape[0]
    R = torch.empty(B, 3, 3, dtype=torch.float32, device=rot.device)
    if B:
        hip.check(lib.catre_rot_to_mat(hip.ptr(rot), int(rot_type_id), hip.ptr(R), B, hip.stream_ptr(rot.device)),
                  "catre_rot_to_mat")
    return R


def rot_to_mat_bwd(rot, rot_type_id, grad_R):
    lib = hip.load()
    rot = rot.contiguous()
    grad_R = hip.require_dev_f32(grad_R.contiguous(), "grad_R", (rot.shape[0], 3, 3))
    g = torch.empty_like(rot)
    if rot.shape[0]:
        hip.check(lib.catre_rot_to_mat_bwd(hip.ptr(rot), int(rot_type_id), hip.ptr(grad_R), hip.ptr(g), rot.shape[0],
                                           hip.stream_ptr(rot.device)), "catre_rot_to_mat_bwd")
    return g
