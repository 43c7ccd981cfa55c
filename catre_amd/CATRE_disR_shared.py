"""``CATRE_disR_shared`` - the drop-in model module of the MI355X-native CATRE hot path.

Same module name, class name, constructor, ``forward`` signature, return types, sub-module and
``state_dict`` names and ``build_model_optimizer(cfg, is_test)`` factory as the reference's
``core/catre/models/CATRE_disR_shared.py`` (forward ``:40-166``, factory ``:291-350``), so
``eval(cfg.MODEL.CATRE.NAME).build_model_optimizer(cfg, ...)`` in the reference's
``main_catre.py:138`` resolves to this file unchanged.

``forward`` launches the hand-written gfx950 kernels of ``libcatre_hip.so`` through the C ABI
(``catre_refine_iter``).  There is no PyTorch-op or CPU fallback: on a CPU tensor, or without the
built library, it raises.  ``refine`` is the fused K-iteration entry point (pose-apply + forward
looped on the device, no Python between iterations).
"""
import copy
import logging

import torch
import torch.nn as nn

from . import hip
from .model_utils import get_rot_head, get_ts_head
from .net_factory import PCLNETS
from .runtime import HipRuntime, opts_from_cfg

logger = logging.getLogger(__name__)


class CATRE_disR_shared(nn.Module):
    def __init__(self, cfg, pcl_net, rot_head, ts_head):
        super().__init__()
        assert cfg.MODEL.CATRE.NAME == "CATRE_disR_shared", cfg.MODEL.CATRE.NAME
        self.cfg = cfg
        self.pcl_net = pcl_net
        self.rot_head = rot_head
        self.ts_head = ts_head
        if getattr(pcl_net, "global_feat", False):
            raise ValueError("CATRE_disR_shared needs per-point features: PCLNET.INIT_CFG.global_feat must be False")
        self._opts = opts_from_cfg(cfg, feature_transform=pcl_net.feature_transform)
        rot_w = 2 * int(getattr(rot_head, "rot_dim", 3))
        if rot_w != hip.ROT_DIMS[self._opts.rot_type]:
            # the reference fails at the first forward instead: get_rot_mat asserts / raises on the width
            # (pose_utils.py:357, lie_algebra.py:23) - its two-head ConvOutPerRotHead emits 2 x rot_dim values, so only
            # rot6d (rot_dim=3) and quat (rot_dim=2) can reach get_rot_mat through it
            raise ValueError(
                f"ROT_TYPE={cfg.MODEL.CATRE.ROT_HEAD.ROT_TYPE!r} needs a {hip.ROT_DIMS[self._opts.rot_type]}-wide rotation "
                f"residual, the rot head emits 2 x rot_dim = {rot_w}"
            )
        if int(ts_head.in_dim) != int(self._opts.ts_in_dim):
            raise ValueError(
                f"TS_HEAD.INIT_CFG.in_dim={ts_head.in_dim} does not match the gathered feature width "
                f"{self._opts.ts_in_dim} implied by WITH_KPS_FEATURE / WITH_INIT_SCALE / WITH_INIT_TRANS"
            )
        self._rt = None
        self._opts_bf16 = type(self._opts).from_buffer_copy(self._opts)
        self._opts_bf16.compute_dtype = hip.DTYPE_BF16
        self._opts_split = type(self._opts).from_buffer_copy(self._opts)
        self._opts_split.compute_dtype = hip.DTYPE_SPLIT

    def _inference_opts(self):
        """fp32 kernels unless reduced precision is requested the way the reference requests it - by running
        the forward under ``torch.cuda.amp.autocast`` (``engine.py:304``, ``TEST.AMP_TEST`` in
        ``catre_evaluator.py``) - or explicitly with ``cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"``.  Reduced precision
        means bf16 GEMM operands with fp32 accumulation / GroupNorm statistics / SO(3) update (an fp16 autocast
        request maps to the same kernels)."""
        want = self.cfg.MODEL.CATRE.get("COMPUTE_DTYPE", None)
        if want is None:
            from .train_ops import autocast_on

            return self._opts_bf16 if autocast_on() else self._opts
        if want in ("bf16", "bfloat16"):
            return self._opts_bf16
        if want in ("fp32", "float32"):
            return self._opts
        if want == "split":
            return self._opts_split
        raise ValueError(f"MODEL.CATRE.COMPUTE_DTYPE={want!r}: expected 'fp32', 'split' or 'bf16'")

    # -- runtime is per-instance state that must never be shared by copies of the module
    def __getstate__(self):
        d = self.__dict__.copy()
        d["_rt"] = None
        return d

    def _runtime(self):
        if self._rt is None:
            num_points = self.rot_head.num_points
            n = int(self.cfg.INPUT.get("NUM_PCL", num_points // 2))
            self._rt = HipRuntime(lambda: dict(self.named_parameters()), n, num_points - n, self._opts.ts_in_dim, root=self)
        return self._rt

    def _named_live_params(self):
        """`dict(self.named_parameters())` without walking the module tree on every forward (0.25 ms of a 4 ms host
        iteration): (name, owning module's `_parameters` dict, key) slots, rebuilt when a parent -> child link of the tree
        or a module's parameter count changes; a re-assigned parameter is seen because the lookup goes through the dict."""
        c = self.__dict__.get("_np_cache")
        if c is None or not (all(d.get(n) is m for d, n, m in c[0]) and all(len(d) == k for d, k in c[1])):
            links = [(parent._modules, n, m) for parent in self.modules() for n, m in parent._modules.items() if m is not None]
            mods = list(self.named_modules())
            counts = [(m._parameters, len(m._parameters)) for _, m in mods]
            slots = [(f"{prefix}.{pn}" if prefix else pn, m._parameters, pn) for prefix, m in mods for pn in m._parameters]
            c = (links, counts, slots)
            self.__dict__["_np_cache"] = c
        return {name: d[pn] for name, d, pn in c[2] if d[pn] is not None}

    def forward(
        self,
        x,
        tfd_kps,
        init_pose,
        init_scale,
        K_zoom=None,
        obj_class=None,
        gt_ego_rot=None,
        gt_trans=None,
        gt_scale=None,
        obj_kps=None,
        mean_scales=None,
        sym_info=None,
        do_loss=False,
        cur_iter=0,
    ):
        """x [B,3,N], tfd_kps [B,3,M] (any strides), init_pose [B,3,4], init_scale [B,3], K_zoom [B,3,3]
        -> ``{"pose_{cur_iter}": [B,3,4], "scale_{cur_iter}": [B,3]}`` (reference ``:122-124``)."""
        live = self._named_live_params()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in live.values())
        if x.shape[0] == 0 and not do_loss:
            # an empty batch (the evaluator skips those, catre_evaluator.py:280-281): nothing to launch
            return {f"pose_{cur_iter}": init_pose.new_zeros(0, 3, 4), f"scale_{cur_iter}": init_scale.new_zeros(0, 3)}
        if x.is_cuda and x.device.index != torch.cuda.current_device():
            # kernels are launched on the stream of the tensors' device: make it current, like a torch op would
            with torch.cuda.device(x.device):
                return self.forward(x, tfd_kps, init_pose, init_scale, K_zoom, obj_class, gt_ego_rot, gt_trans, gt_scale,
                                    obj_kps, mean_scales, sym_info, do_loss, cur_iter)
        if not (do_loss or needs_grad):
            # inference: the fused kernels (one launch chain, nothing saved)
            pose, scale = self._runtime().refine_iter(x, tfd_kps, init_pose, init_scale, K_zoom, mean_scales,
                                                      self._inference_opts())
            return {f"pose_{cur_iter}": pose, f"scale_{cur_iter}": scale}

        # training / autograd: layer-by-layer HIP ops chained by torch.autograd (catre_amd/train_forward.py).
        # Like the reference, gradients flow to the parameters only: the caller detaches the fed-back pose
        # (engine.py:324-325) and x / tfd_kps come from the data batch.
        from .train_forward import forward_train
        from .train_ops import amp_mode, train_kernels

        # under torch.autocast (SOLVER.AMP.ENABLED in the reference's loop) the row GEMMs take bf16 operands;
        # cfg.MODEL.CATRE.COMPUTE_DTYPE forces either precision; cfg.MODEL.CATRE.TRAIN_KERNELS picks kernel forms (A/B)
        with amp_mode(self.cfg.MODEL.CATRE.get("COMPUTE_DTYPE", None)), \
                train_kernels(self.cfg.MODEL.CATRE.get("TRAIN_KERNELS", None)):
            pose, scale, aux = forward_train(live, self._opts, x, tfd_kps, init_pose, init_scale, K_zoom, mean_scales,
                                             rt=self._runtime())
        out_dict = {f"pose_{cur_iter}": pose, f"scale_{cur_iter}": scale}
        if not do_loss:
            return out_dict
        assert gt_ego_rot is not None and (gt_trans is not None)
        from .losses import VisScalars, catre_loss

        loss_dict, vis = catre_loss(self.cfg, out_rot=pose[:, :3, :3], out_trans=pose[:, :3, 3], out_scale=scale,
                                    gt_rot=gt_ego_rot, gt_trans=gt_trans, gt_scale=gt_scale, obj_kps=obj_kps,
                                    sym_info=sym_info, trans_deltas=aux["trans_deltas"], return_vis=True, pose=pose)
        # The reference pushes 14 `.item()` scalars per call into detectron2's EventStorage (:127-164: vis/error_R,
        # vis/error_t, object 0's translation / deltas / ground truth).  Here the loss kernels write them into ONE device
        # tensor; `self.vis_scalars` exposes it (no copy until read) and, when an EventStorage is active, the same keys
        # are written with a single 14-float copy - never while the step is being captured into a HIP graph.
        self.vis_scalars = VisScalars(vis, cur_iter)
        storage = _active_event_storage()
        if storage is not None and self.cfg.MODEL.CATRE.get("LOG_VIS_SCALARS", True) \
                and not torch.cuda.is_current_stream_capturing():
            storage.put_scalars(**self.vis_scalars.as_dict())
        return out_dict, loss_dict

    def catre_loss(self, out_rot, out_trans, out_scale, gt_rot=None, gt_trans=None, gt_scale=None, obj_kps=None,
                   sym_info=None):
        """Same signature as the reference method (``:168-178``)."""
        from .losses import catre_loss

        return catre_loss(self.cfg, out_rot, out_trans, out_scale, gt_rot, gt_trans, gt_scale, obj_kps, sym_info)

    @torch.no_grad()
    def refine(self, batch, n_iter=None):
        """The whole test-time loop of ``catre_inference_on_dataset`` (reference
        ``core/catre/engine/catre_evaluator.py:292-311``) as one stream of kernel launches.

        ``batch`` needs ``pcl [B,N,3]``, ``obj_kps [B,M,3]``, ``obj_pose_est [B,3,4]``, ``obj_scale_est [B,3]``,
        ``K [B,3,3]`` and (for "mean" scale types) ``obj_mean_scales [B,3]``.  Returns the reference's
        ``out_dict``: ``pose_0..pose_K`` / ``scale_0..scale_K``.
        """
        n_iter = int(self.cfg.MODEL.CATRE.N_ITER_TEST if n_iter is None else n_iter)
        if batch["pcl"].shape[0] == 0:
            out = {}
            for i in range(n_iter + 1):
                out[f"pose_{i}"] = batch["obj_pose_est"].new_zeros(0, 3, 4)
                out[f"scale_{i}"] = batch["obj_scale_est"].new_zeros(0, 3)
            return out
        with torch.cuda.device(batch["pcl"].device if batch["pcl"].is_cuda else None):
            poses, scales = self._runtime().refine_k(
                batch["pcl"], batch["obj_kps"], batch["obj_pose_est"], batch["obj_scale_est"], batch.get("K"),
                batch.get("obj_mean_scales"), self._inference_opts(), n_iter,
            )
        out = {}
        for i in range(n_iter + 1):
            out[f"pose_{i}"], out[f"scale_{i}"] = poses[i], scales[i]
        return out


def _active_event_storage():
    """detectron2's current ``EventStorage`` if the caller opened one (``engine.py:266``), else None (detectron2 absent or
    no storage context: nothing to log into)."""
    try:
        from detectron2.utils.events import get_event_storage

        return get_event_storage()
    except Exception:  # ImportError, or detectron2's AssertionError outside a `with EventStorage(...)` block
        return None


def _maybe_add_gradient_clipping(cfg, optimizer):
    """``lib/torch_utils/solver/grad_clip_d2.py:80-120``: with ``SOLVER.CLIP_GRADIENTS.ENABLED`` the optimizer's class is
    replaced by a subclass whose ``step`` clips first - ``CLIP_TYPE`` "value" / "norm" per parameter, "full_model" (the
    default) over all parameters at once."""
    clip = cfg.SOLVER.get("CLIP_GRADIENTS", None)
    if not clip or not clip.get("ENABLED", False):
        return optimizer
    clip_value, norm_type = float(clip.get("CLIP_VALUE", 1.0)), float(clip.get("NORM_TYPE", 2.0))
    clip_type = clip.get("CLIP_TYPE", "full_model")
    if clip_type not in ("value", "norm", "full_model"):
        raise ValueError(f"'{clip_type}' is not a valid GradientClipType")
    base = type(optimizer)

    def step(self, closure=None):
        groups = [[p for p in g["params"] if p.grad is not None] for g in self.param_groups]
        if clip_type == "full_model":
            torch.nn.utils.clip_grad_norm_([p for g in groups for p in g], clip_value, norm_type)
        else:
            for p in (p for g in groups for p in g):
                if clip_type == "value":
                    torch.nn.utils.clip_grad_value_(p, clip_value)
                else:
                    torch.nn.utils.clip_grad_norm_(p, clip_value, norm_type)
        return base.step(self, closure)

    optimizer.__class__ = type(base.__name__ + "WithGradientClip", (base,), {"step": step})
    return optimizer


def _build_optimizer(cfg, params_lr_list):
    """``core/utils/solver_utils.build_optimizer_with_params`` (reference ``:75-87``): ``OPTIMIZER_CFG`` (a dict, or the
    string form the reference ``eval``s) names the optimizer and carries its keyword arguments; the shipped config's
    Ranger is ``catre_amd.ranger.Ranger`` (one fused multi-tensor HIP step); ``torch.optim`` types get every
    keyword; then ``maybe_add_gradient_clipping``."""
    ocfg = cfg.SOLVER.get("OPTIMIZER_CFG", "")
    if isinstance(ocfg, str):
        if ocfg == "":
            raise RuntimeError("please provide cfg.SOLVER.OPTIMIZER_CFG to build optimizer")  # solver_utils.py:77
        ocfg = eval(ocfg, {"__builtins__": {}}, {"dict": dict})  # "dict(type='Ranger', lr=1e-4, ...)" (:79)
    ocfg = dict(ocfg)
    typ = ocfg.pop("type")
    groups = [dict(params=list(g["params"]), lr=g["lr"]) for g in params_lr_list]
    if typ == "Ranger":  # the shipped config (…_120e.py:49)
        from .ranger import Ranger

        opt = Ranger(groups, **ocfg)
    elif hasattr(torch.optim, typ):
        opt = getattr(torch.optim, typ)(groups, **ocfg)
    else:
        # Ranger21 / Lamb / MADGRAD / NAdamW / AdaBelief / SGDP / AdamP / SGD_GC of solver_utils.py:30-72 are generic
        # third-party optimizers outside the hot path
        raise ValueError(f"Unknown optimizer name: {typ} (available: 'Ranger' and every torch.optim class)")
    return _maybe_add_gradient_clipping(cfg, opt)


def _load_pretrained_pcl_net(model, path):
    """``PCLNET.PRETRAINED`` = a checkpoint file (reference ``:342-346``, mmcv ``load_checkpoint(..., strict=False)``)."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt.get("state_dict", ckpt.get("model", ckpt)) if isinstance(ckpt, dict) else ckpt
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    missing, unexpected = model.pcl_net.load_state_dict(sd, strict=False)
    logger.info("load pcl_net weights from: %s (missing %d, unexpected %d keys)", path, len(missing), len(unexpected))


def build_model_optimizer(cfg, is_test=False):
    """reference ``CATRE_disR_shared.py:291-350``: returns ``(model, optimizer-or-None)``; param groups are
    pcl_net @ BASE_LR, rot_head / ts_head @ BASE_LR x LR_MULT; ends with ``model.to(cfg.MODEL.DEVICE)``."""
    pcl_net_cfg = cfg.MODEL.CATRE.PCLNET
    params_lr_list = []
    init_pcl_net_args = dict(copy.deepcopy(pcl_net_cfg.INIT_CFG))
    pcl_net_type = init_pcl_net_args.pop("type")
    pcl_net = PCLNETS[pcl_net_type](**init_pcl_net_args)
    if pcl_net_cfg.get("FREEZE", False):
        for param in pcl_net.parameters():  # (the reference calls .parameters() on the cfg dict here: a latent bug)
            param.requires_grad = False
    else:
        params_lr_list.append(
            {"params": filter(lambda p: p.requires_grad, pcl_net.parameters()), "lr": float(cfg.SOLVER.BASE_LR)}
        )
    rot_head, rot_head_params = get_rot_head(cfg)
    params_lr_list.extend(rot_head_params)
    ts_head, ts_head_params = get_ts_head(cfg)
    params_lr_list.extend(ts_head_params)

    model = CATRE_disR_shared(cfg, pcl_net, rot_head, ts_head)
    optimizer = None if is_test else _build_optimizer(cfg, params_lr_list)
    if cfg.MODEL.get("WEIGHTS", "") == "":  # reference :328-346
        pretrained = pcl_net_cfg.get("PRETRAINED", "")
        if pretrained == "":
            logger.warning("Randomly initialize weights for pcl_net!")
        elif pretrained in ("timm", "internal"):
            logger.info("Check if the pcl_net has been initialized with its own method!")
        else:
            _load_pretrained_pcl_net(model, pretrained)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model, optimizer


def expected_state_shapes(cfg):
    """``{state_dict key: shape}`` of the model ``cfg`` describes (SURVEY.md section 8b listing)."""
    net = cfg.MODEL.CATRE
    P = int(net.ROT_HEAD.INIT_CFG.num_points)
    rd = int(net.ROT_HEAD.INIT_CFG.get("rot_dim", 3))
    ts_in = int(net.TS_HEAD.INIT_CFG.in_dim)
    ft = bool(net.PCLNET.INIT_CFG.get("feature_transform", False))
    s = {}

    def stn(prefix, k):
        for name, o, i in (("conv1", 64, k), ("conv2", 128, 64), ("conv3", 1024, 128)):
            s[f"{prefix}.{name}.weight"], s[f"{prefix}.{name}.bias"] = (o, i, 1), (o,)
        for name, o, i in (("fc1", 512, 1024), ("fc2", 256, 512), ("fc3", k * k, 256)):
            s[f"{prefix}.{name}.weight"], s[f"{prefix}.{name}.bias"] = (o, i), (o,)

    stn("pcl_net.stn", 3)
    for name, o, i in (("conv1", 64, 3), ("conv2", 128, 64), ("conv3", 512, 128), ("conv4", 1024, 512)):
        s[f"pcl_net.{name}.weight"], s[f"pcl_net.{name}.bias"] = (o, i, 1), (o,)
    if ft:
        stn("pcl_net.fstn", 64)
    for a in ("x", "y"):
        p = f"rot_head.rot_head_{a}"
        s[f"{p}.norm.weight"], s[f"{p}.norm.bias"] = (256,), (256,)
        s[f"{p}.layers.0.weight"], s[f"{p}.layers.0.bias"] = (256, 1088, 1), (256,)
        s[f"{p}.layers.1.weight"], s[f"{p}.layers.1.bias"] = (256,), (256,)
        s[f"{p}.layers.3.weight"], s[f"{p}.layers.3.bias"] = (256, 256, 1), (256,)
        s[f"{p}.layers.4.weight"], s[f"{p}.layers.4.bias"] = (256,), (256,)
        s[f"{p}.neck.0.weight"], s[f"{p}.neck.0.bias"] = (rd, 256, 1), (rd,)
        s[f"{p}.conv_p.weight"], s[f"{p}.conv_p.bias"] = (1, P, 1), (1,)
    s["ts_head.norm.weight"], s["ts_head.norm.bias"] = (256,), (256,)
    s["ts_head.linears.0.weight"], s["ts_head.linears.0.bias"] = (256, ts_in), (256,)
    s["ts_head.linears.1.weight"], s["ts_head.linears.1.bias"] = (256,), (256,)
    s["ts_head.linears.3.weight"], s["ts_head.linears.3.bias"] = (256, 256), (256,)
    s["ts_head.linears.4.weight"], s["ts_head.linears.4.bias"] = (256,), (256,)
    for n in ("fc_t", "fc_s"):
        s[f"ts_head.{n}.weight"], s[f"ts_head.{n}.bias"] = (3, 256), (3,)
    return s
