"""Generate ``tests/golden/*.npz`` by running the REFERENCE ITSELF (imported from
``/root/reference`` through ``oracle/ref_shim.py``) on seeded synthetic inputs.

Runs only in the build container (the reference does not exist on the GPU box):

    python -m oracle.make_golden            # rewrites tests/golden/*.npz

Each fixture stores the inputs (so nothing depends on RNG reproducibility), the name of the
weight recipe (``catre_amd.synth.recipe_state_dict`` - regenerated, not stored: 17 MB) and the
reference outputs: ``pose_i`` / ``scale_i`` for every refine iteration plus per-stage
intermediates of iteration 1 captured with forward hooks on the reference sub-modules.

ORACLE TOOLING - never imported by the product.
"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from catre_amd import synth  # noqa: E402
from oracle import ref_shim  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().float().cpu().numpy().copy()


def reference_cfg(N, M, overrides=None):
    cfg = ref_shim.load_reference_cfg()
    cfg.MODEL.DEVICE = "cpu"
    cfg.INPUT.NUM_PCL, cfg.INPUT.NUM_KPS = N, M
    cfg.MODEL.CATRE.PCLNET.INIT_CFG.num_points = N
    cfg.MODEL.CATRE.ROT_HEAD.INIT_CFG.num_points = N + M
    for path, v in (overrides or {}).items():
        node = cfg
        keys = path.split(".")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = v
    return cfg


def run_reference(cfg, batch, n_iter, salt=0, amp_dtype=None):
    """The loop of catre_evaluator.py:292-311 with the reference's own batch_updater_test + model; with
    ``amp_dtype`` the forward runs under autocast like ``catre_evaluator.py``'s ``amp_test`` branch."""
    ref_shim.install()
    from core.catre.engine.batch_test import batch_updater_test

    model = ref_shim.build_reference_model(cfg).eval()
    sd = synth.recipe_state_dict({k: v.shape for k, v in model.state_dict().items()}, salt)
    model.load_state_dict(sd, strict=True)

    cap = {}
    calls = {"pcl": 0}

    def hook_pcl(mod, inp, out):
        tag = "x" if calls["pcl"] % 2 == 0 else "k"
        if calls["pcl"] < 2:
            cap[f"g_{tag}"] = _np(out[:, :1024, 0])
            cap[f"pointfeat_{tag}"] = _np(out[:, 1024:, :64])  # first 64 points
            cap[f"pointfeat_max_{tag}"] = _np(out[:, 1024:, :].max(2)[0])
        calls["pcl"] += 1

    def mk(name, counter):
        def hook(mod, inp, out):
            i = counter.setdefault(name, 0)
            if i < 2:
                cap[f"{name}_{'x' if i == 0 else 'k'}"] = _np(out)
            counter[name] = i + 1
        return hook

    def hook_rot(mod, inp, out):  # hooks must return None (a value would replace the output)
        cap.setdefault("rot_deltas", _np(out))

    def hook_ts(mod, inp, out):
        cap.setdefault("trans_deltas", _np(out[0]))
        cap.setdefault("scale_deltas", _np(out[1]))

    ctr = {}
    hs = [
        model.pcl_net.register_forward_hook(hook_pcl),
        model.pcl_net.stn.register_forward_hook(mk("trans", ctr)),
        model.rot_head.register_forward_hook(hook_rot),
        model.ts_head.register_forward_hook(hook_ts),
    ]
    if hasattr(model.pcl_net, "fstn"):
        hs.append(model.pcl_net.fstn.register_forward_hook(mk("transfeat", ctr)))

    b = {k: (v.clone() if isinstance(v, torch.Tensor) else copy.deepcopy(v)) for k, v in batch.items()}
    out = {"pose_0": _np(b["obj_pose_est"]), "scale_0": _np(b["obj_scale_est"])}
    poses_est, scales_est = None, None
    with torch.no_grad():
        for i in range(1, n_iter + 1):
            batch_updater_test(cfg, b, poses_est=poses_est, scales_est=scales_est, device="cpu")
            if i == 1:
                cap["x_in"] = _np(b["x"][:, :, :64])
                cap["tfd_kps_in"] = _np(b["tfd_kps"][:, :, :64])
            with torch.autocast("cpu", dtype=amp_dtype, enabled=amp_dtype is not None):
                o = model(
                    b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"],
                    K_zoom=b["K"], obj_class=b["obj_cls"], mean_scales=b["obj_mean_scales"],
                    do_loss=False, cur_iter=i,
                )
            o = {k: v.float() for k, v in o.items()}
            poses_est, scales_est = o[f"pose_{i}"], o[f"scale_{i}"]
            out[f"pose_{i}"], out[f"scale_{i}"] = _np(poses_est), _np(scales_est)
    for h in hs:
        h.remove()
    out.update({f"stage_{k}": v for k, v in cap.items()})
    return out


CASES = {
    # name: (B, N, M, K, seed, salt, prior, cfg overrides)
    "refine_b2_n1024": (2, 1024, 1024, 4, 0, 0, None, {}),
    "refine_b3_ragged": (3, 1000, 500, 2, 1, 1, None, {}),
    "refine_b1_bottle": (1, 1024, 1024, 4, 2, 0, "bottle", {}),
    "refine_b2_small": (2, 100, 37, 2, 3, 2, None, {}),
    "refine_b2_3d_mul": (2, 256, 256, 2, 4, 0, None, {
        "MODEL.CATRE.ROT_HEAD.DELTA_T_SPACE": "3D", "MODEL.CATRE.ROT_HEAD.SCLAE_TYPE": "mean_mul"}),
    "refine_b2_deepim_noK": (2, 256, 256, 2, 5, 0, None, {
        "MODEL.CATRE.ROT_HEAD.DELTA_Z_STYLE": "deepim", "MODEL.CATRE.ROT_HEAD.T_TRANSFORM_K_AWARE": False,
        "MODEL.CATRE.ROT_HEAD.DELTA_T_WEIGHT": 0.1}),
    "refine_b2_allo": (2, 256, 256, 2, 6, 0, None, {"MODEL.CATRE.ROT_HEAD.ROT_TYPE": "allo_rot6d"}),
    "refine_b2_norefscale_nozc": (2, 256, 256, 2, 7, 0, None, {
        "MODEL.REFINE_SCLAE": False, "INPUT.ZERO_CENTER_INPUT": False}),
    # BASELINE.json config 5 shape (N=2048 observed, M=1024 prior, K=8 deep iteration), fp32
    "refine_b1_n2048_k8": (1, 2048, 1024, 8, 9, 0, None, {}),
    # PointNet without the feature transform (PCLNET.INIT_CFG.feature_transform=False, pointnet.py:105)
    "refine_b2_noft": (2, 192, 128, 2, 10, 0, None, {"MODEL.CATRE.PCLNET.INIT_CFG.feature_transform": False}),
    # quaternion residual: two rot heads of width rot_dim=2 -> [B,4] -> quat2mat_torch (model_utils.py:29-30)
    "refine_b2_quat": (2, 256, 128, 3, 12, 0, None, {
        "MODEL.CATRE.ROT_HEAD.ROT_TYPE": "ego_quat", "MODEL.CATRE.ROT_HEAD.INIT_CFG.rot_dim": 2}),
    "refine_b2_allo_quat": (2, 128, 256, 2, 13, 1, None, {
        "MODEL.CATRE.ROT_HEAD.ROT_TYPE": "allo_quat", "MODEL.CATRE.ROT_HEAD.INIT_CFG.rot_dim": 2}),
    "refine_b2_kpsfeat_trans": (2, 256, 256, 2, 8, 0, None, {
        "MODEL.CATRE.TS_HEAD.WITH_KPS_FEATURE": True, "MODEL.CATRE.TS_HEAD.WITH_INIT_TRANS": True,
        "MODEL.CATRE.TS_HEAD.INIT_CFG.in_dim": 1088 * 2 + 3 + 3}),
}


TRAIN_CASES = {
    # name: (B, N, M, seed, salt, symmetric object indices, #sym rotations)
    "train_b4": (4, 128, 96, 21, 0, (1, 3), 12),
    # N and M multiples of 64: the shape class whose training forward takes the fused kernels (and, under autocast, the
    # one-node bf16-row rotation heads)
    "train_b4_t64": (4, 128, 64, 22, 0, (0, 2), 12),
}


def train_sym_info(B, sym_idx, nsym):
    from oracle.catre_oracle import y_axis_symmetries

    return [y_axis_symmetries(nsym) if i in sym_idx else None for i in range(B)]


def grad_sample_index(numel, n=512):
    """The entries of a flattened gradient the AMP training fixture keeps: every one up to n, else n evenly spaced."""
    if numel <= n:
        return np.arange(numel, dtype=np.int64)
    return np.unique(np.linspace(0, numel - 1, n).astype(np.int64))


def run_reference_train(cfg, batch, sym_info, salt=0, amp_dtype=None, samples=False):
    """One training iteration of the reference (engine.py:293-349 without the optimizer): batch_updater_test
    pose-apply, model(..., do_loss=True), sum of the loss dict, backward.  Records the losses and, per parameter,
    the gradient L2 norm and its first 64 entries.  With ``amp_dtype`` the forward runs inside
    ``torch.autocast("cpu", dtype=amp_dtype)`` like engine.py:304's AMP branch (no GradScaler: bf16 needs none) and
    every gradient additionally leaves ``gradsample__*``: its entries at ``grad_sample_index`` (``samples``: those for the
    fp32 run as well)."""
    ref_shim.install()
    from core.catre.engine.batch_test import batch_updater_test

    model = ref_shim.build_reference_model(cfg).train()
    sd = synth.recipe_state_dict({k: v.shape for k, v in model.state_dict().items()}, salt)
    model.load_state_dict(sd, strict=True)
    b = {k: (v.clone() if isinstance(v, torch.Tensor) else copy.deepcopy(v)) for k, v in batch.items()}
    batch_updater_test(cfg, b, device="cpu")
    # the scalars the reference's forward pushes into detectron2's EventStorage (CATRE_disR_shared.py:127-164)
    import core.catre.models.CATRE_disR_shared as ref_mod

    logged = {}

    class _Storage:
        def put_scalars(self, **kw):
            logged.update(kw)

    saved = ref_mod.get_event_storage
    ref_mod.get_event_storage = lambda: _Storage()
    import contextlib

    # numpy has no bfloat16: the reference's logging / symmetry helpers call `.cpu().numpy()` on network outputs
    # (model_utils.py:228, pose_utils.py get_closest_rot_batch), which raises for bf16 tensors on any device.  For the
    # autocast run only, Tensor.numpy upcasts bf16 first (exact) - the reference source itself stays unmodified.
    orig_numpy = torch.Tensor.numpy
    if amp_dtype is torch.bfloat16:
        torch.Tensor.numpy = lambda self, *a, **k: orig_numpy(self.float() if self.dtype == torch.bfloat16 else self, *a, **k)
    try:
        with (torch.autocast("cpu", dtype=amp_dtype) if amp_dtype is not None else contextlib.nullcontext()):
            out_dict, loss_dict = model(
                b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                obj_class=b["obj_cls"], gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"],
                obj_kps=b["obj_kps"], mean_scales=b["obj_mean_scales"], sym_info=sym_info, do_loss=True, cur_iter=1,
            )
            losses = sum(loss_dict.values())
    finally:
        ref_mod.get_event_storage = saved
        torch.Tensor.numpy = orig_numpy
    losses.backward()
    out = {"pose_1": _np(out_dict["pose_1"]), "scale_1": _np(out_dict["scale_1"])}
    assert len(logged) == 14, sorted(logged)
    for k, v in logged.items():
        out["vis__" + k.replace("/", "__")] = np.array([float(v)], dtype=np.float64)
    for k, v in loss_dict.items():
        out[f"loss__{k}"] = _np(v.reshape(1))
    for k, p in model.named_parameters():
        if p.grad is None:
            out[f"gradnone__{k}"] = np.zeros(1, dtype=np.float32)
        else:
            out[f"gradnorm__{k}"] = _np(p.grad.norm().reshape(1))
            out[f"gradhead__{k}"] = _np(p.grad.reshape(-1)[:64])
            if amp_dtype is not None or samples:
                out[f"gradsample__{k}"] = _np(p.grad.reshape(-1)[torch.from_numpy(grad_sample_index(p.grad.numel()))])
    return out


def make_train_golden(name):
    B, N, M, seed, salt, sym_idx, nsym = TRAIN_CASES[name]
    batch = synth.make_inputs(B, N, M, seed=seed)
    cfg = reference_cfg(N, M, {})
    out = run_reference_train(cfg, batch, train_sym_info(B, sym_idx, nsym), salt)
    arrays = {f"in_{k}": _np(v) for k, v in batch.items()}
    arrays.update(out)
    arrays["meta"] = np.array([B, N, M, 1, seed, salt], dtype=np.int64)
    arrays["meta_sym"] = np.array(list(sym_idx) + [nsym], dtype=np.int64)
    arrays["meta_overrides"] = np.array(repr([]))
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
    print("   losses:", {k[6:]: float(v[0]) for k, v in out.items() if k.startswith("loss__")})


def make_amp_train_golden(name="train_b4"):
    """engine.py:304,333-347 on the REFERENCE: the ``train_b4`` iteration again, forward under bf16 autocast.  Stores the
    losses, pose / scale, and per parameter the gradient norm, head and 512 sampled entries - plus the same samples of the
    reference's fp32 gradients (so a test can ask "no further from fp32 than the reference's own autocast is")."""
    B, N, M, seed, salt, sym_idx, nsym = TRAIN_CASES[name]
    batch = synth.make_inputs(B, N, M, seed=seed)
    cfg = reference_cfg(N, M, {})
    sym = train_sym_info(B, sym_idx, nsym)
    out = run_reference_train(cfg, batch, sym, salt, amp_dtype=torch.bfloat16)
    ref32 = run_reference_train(cfg, batch, sym, salt, samples=True)
    arrays = dict(out)
    for k, v in ref32.items():
        if k.startswith("gradsample__") or k.startswith("loss__"):
            arrays["fp32__" + k] = v
    arrays["meta"] = np.array([B, N, M, 1, seed, salt], dtype=np.int64)
    arrays["meta_sym"] = np.array(list(sym_idx) + [nsym], dtype=np.int64)
    arrays["meta_inputs"] = np.array(f"{name}.npz")
    path = os.path.join(GOLDEN_DIR, f"amp_{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"amp_{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
    print("   losses bf16:", {k[6:]: float(v[0]) for k, v in out.items() if k.startswith("loss__")})
    print("   losses fp32:", {k[6:]: float(v[0]) for k, v in ref32.items() if k.startswith("loss__")})


AMP_CASES = ("refine_b2_n1024", "refine_b1_n2048_k8")


def make_amp_golden():
    """The reference under bf16 autocast on the inputs of two fp32 cases: only pose_i / scale_i are kept (the
    inputs live in the fp32 fixture).  Used to characterise what 'reduced precision' costs the REFERENCE."""
    out = {}
    for name in AMP_CASES:
        B, N, M, K, seed, salt, prior, ov = CASES[name]
        batch = synth.make_inputs(B, N, M, seed=seed)
        r = run_reference(reference_cfg(N, M, ov), batch, K, salt, amp_dtype=torch.bfloat16)
        for i in range(K + 1):
            out[f"{name}__pose_{i}"], out[f"{name}__scale_{i}"] = r[f"pose_{i}"], r[f"scale_{i}"]
    path = os.path.join(GOLDEN_DIR, "amp_bf16_reference.npz")
    np.savez_compressed(path, **out)
    print(f"amp: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def make_aug_golden(seed=77):
    """Reference aug_3d_bbox / aug_RT / aug_poses_normal / aug_scale_normal (CPU) with the random draws replayed
    and stored, so that the HIP kernels (which take the draws as inputs) can be checked against these outputs."""
    ref_shim.install()
    import random

    from core.catre.engine import engine_utils as EU
    from core.utils import pose_aug as PA

    B, N = 6, 160
    inp = synth.make_inputs(B, N, 32, seed=seed)
    sym_flags = [0, 1, 0, 1, 1, 0]
    batch = {"pcl": inp["pcl"].clone(), "obj_pose": torch.cat([inp["gt_rot"], inp["gt_trans"].unsqueeze(-1)], -1),
             "obj_scale": inp["gt_scale"].clone(), "sym_info": [np.eye(3)[None] if f else None for f in sym_flags]}
    out = {"in_pcl": _np(batch["pcl"]), "in_pose": _np(batch["obj_pose"]), "in_scale": _np(batch["obj_scale"]),
           "in_sym": np.array(sym_flags, dtype=np.int32)}

    def reseed():
        torch.manual_seed(seed)
        np.random.seed(seed)
        random.seed(seed)

    reseed()
    EU.aug_3d_bbox(batch, device="cpu")
    reseed()
    ex, ey, ez = torch.rand(3)
    out["bbox_ratios"] = np.array([ex * 0.4 + 0.8, ey * 0.4 + 0.8, ez * 0.4 + 0.8], dtype=np.float32)
    out["bbox_pcl"], out["bbox_scale"] = _np(batch["pcl"]), _np(batch["obj_scale"])

    reseed()
    EU.aug_RT(batch, device="cpu")
    reseed()
    rx, ry, rz = torch.rand(3) * 15.0 * 2 - 15.0
    tx, ty, tz = torch.rand(1) * 0.005 * 2 - 0.005, torch.rand(1) * 0.005 * 2 - 0.005, torch.rand(1) * 0.025 * 2 - 0.025
    out["rt_delta_r"] = _np(EU.get_rotation_torch(rx, ry, rz))
    out["rt_delta_t"] = np.array([tx, ty, tz], dtype=np.float32).reshape(3)
    out["rt_pcl"], out["rt_pose"] = _np(batch["pcl"]), _np(batch["obj_pose"])

    std_rot, std_trans, std_scale = (10, 5, 2.5, 1.25), [(0.02, 0.02, 0.02), (0.01, 0.01, 0.01), (0.005, 0.005, 0.005)], \
        [(0.01, 0.01, 0.01), (0.005, 0.005, 0.005), (0.002, 0.002, 0.002)]
    poses = batch["obj_pose"].clone()
    poses[0, 2, 3] = 0.05  # exercises the min_z clamp
    reseed()
    out["noise_pose_out"] = _np(PA.aug_poses_normal(poses, std_rot=std_rot, std_trans=std_trans, max_rot=1.0, min_z=0.1))
    reseed()
    sr = np.random.choice(std_rot)
    out["noise_euler_deg"] = _np(torch.normal(mean=0, std=sr, size=(B, 3)))
    st = std_trans[np.random.choice(len(std_trans))]
    out["noise_trans"] = _np(torch.normal(mean=torch.zeros(B, 3), std=torch.tensor(st).view(1, 3)))
    out["noise_pose_in"] = _np(poses)
    out["noise_sel"] = np.array([sr, *st], dtype=np.float64)

    scales = batch["obj_scale"].clone()
    scales[1, 0] = 0.041  # exercises the min_s clamp
    reseed()
    out["noise_scale_out"] = _np(PA.aug_scale_normal(scales, std_scale=std_scale, min_s=0.04))
    reseed()
    ss = std_scale[np.random.choice(len(std_scale))]
    out["noise_scale"] = _np(torch.normal(mean=torch.zeros(B, 3), std=torch.tensor(ss).view(1, 3)))
    out["noise_scale_in"] = _np(scales)

    # self-check of the replayed draws against the restatement (fails loudly if the replay is wrong)
    from oracle import aug_oracle as AO

    t = torch.from_numpy
    p1, s1 = AO.aug_3d_bbox(t(out["in_pcl"]), t(out["in_pose"]), t(out["in_scale"]), t(out["in_sym"]), out["bbox_ratios"])
    assert np.abs(_np(p1) - out["bbox_pcl"]).max() < 1e-6 and np.abs(_np(s1) - out["bbox_scale"]).max() < 1e-7
    p2, q2 = AO.aug_rt(p1, t(out["in_pose"]), t(out["rt_delta_r"]), t(out["rt_delta_t"]))
    assert np.abs(_np(p2) - out["rt_pcl"]).max() < 1e-6 and np.abs(_np(q2) - out["rt_pose"]).max() < 1e-6
    pn = AO.poses_from_noise(t(out["noise_pose_in"]), t(out["noise_euler_deg"]), t(out["noise_trans"]), 1.0, 0.1)
    assert np.abs(_np(pn) - out["noise_pose_out"]).max() < 1e-6
    sn = AO.scales_from_noise(t(out["noise_scale_in"]), t(out["noise_scale"]), 0.04, 0.45)
    assert np.abs(_np(sn) - out["noise_scale_out"]).max() < 1e-7
    path = os.path.join(GOLDEN_DIR, "aug_train.npz")
    np.savez_compressed(path, **out)
    print(f"aug: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def make_pcl_golden(seed=11, N=96):
    """Reference crop_ball_from_depth_image / crop_mask_depth_image per instance of a synthetic depth frame, with the
    torch.randperm draws replayed and stored."""
    ref_shim.install()
    from core.utils import cat_data_utils as CU
    from lib.pysixd import misc
    from oracle import pcl_oracle as PO

    sc = synth.make_depth_scene(seed=seed)
    depth, K, masks, poses, scales = sc["depth"], sc["K"], sc["masks"], sc["poses"], sc["scales"]
    H, W = depth.shape
    I = len(masks)
    depth_bp = misc.backproject_th(depth, K.numpy())
    image = torch.zeros(H, W, 3, dtype=torch.uint8)
    out = {f"in_{k}": _np(v) if v.dtype != torch.bool else v.numpy() for k, v in sc.items()}
    out["meta"] = np.array([N, seed], dtype=np.int64)
    for mode, ball in (("ball", True), ("mask", False)):
        torch.manual_seed(seed)
        ref = []
        for i in range(I):
            if ball:
                _, pcl, _ = CU.crop_ball_from_depth_image(image, depth_bp, masks[i], poses[i], scales[i], ratio=0.5,
                                                          cam_intrinsics=K, num_points=N, device="cpu", fps_sample=False)
            else:
                _, pcl, _ = CU.crop_mask_depth_image(image, depth_bp, masks[i], num_points=N)
            ref.append(pcl.to(torch.float32))
        out[f"{mode}_pcl"] = _np(torch.stack(ref))
        torch.manual_seed(seed)
        cnt, sidx = [], []
        for i in range(I):
            pix, bp = PO.candidates(depth, K, masks[i], poses[i], scales[i], 0.5, use_ball=ball)
            cnt.append(len(pix))
            # ball crop: one permutation of the tiled list; mask crop: random_sample, which tops a short list up
            s = torch.randperm(PO.tiled_length(len(pix), N))[:N] if ball else PO.random_sample_idx(len(pix), N)
            sidx.append(s)
            got, _ = PO.sample(pix, bp, s)
            assert np.abs(_np(got) - out[f"{mode}_pcl"][i]).max() < 1e-7, (mode, i)
        out[f"{mode}_counts"] = np.array(cnt, dtype=np.int64)
        out[f"{mode}_sample_idx"] = _np(torch.stack(sidx)).astype(np.int64)
    # INPUT.FPS_SAMPLE: the reference's ball crop with farthest point sampling (deterministic: no random draw)
    fps_pcl, fps_idx = [], []
    for i in range(I):
        _, pcl, _ = CU.crop_ball_from_depth_image(image, depth_bp, masks[i], poses[i], scales[i], ratio=0.5,
                                                  cam_intrinsics=K, num_points=N, device="cpu", fps_sample=True)
        pix, bp = PO.candidates(depth, K, masks[i], poses[i], scales[i], 0.5, use_ball=True)
        s = PO.fps_sample_idx(pix, bp, N)
        got, _ = PO.sample(pix, bp, s)
        assert pcl.shape[0] >= N and np.abs(_np(got) - _np(pcl[:N].to(torch.float32))).max() < 1e-7, ("fps", i)
        fps_pcl.append(pcl[:N].to(torch.float32))
        fps_idx.append(s)
    out["fps_pcl"] = _np(torch.stack(fps_pcl))
    out["fps_sample_idx"] = _np(torch.stack(fps_idx)).astype(np.int64)
    path = os.path.join(GOLDEN_DIR, "pcl_prep.npz")
    np.savez_compressed(path, **out)
    print(f"pcl: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); ball counts {out['ball_counts']}, mask counts {out['mask_counts']}")


RANGER_SHAPES = [(8, 5, 1), (6, 7), (9,), (4, 3, 2, 2), (16, 40)]
RANGER_STEPS = 14


def ranger_problem(seed=31):
    """Seeded parameters and per-step gradients (one NaN / +inf / -inf injected) shared by the golden and the tests."""
    g = torch.Generator().manual_seed(seed)
    params = [torch.randn(s, generator=g) for s in RANGER_SHAPES]
    grads = [[torch.randn(s, generator=g) * (0.5 + 0.1 * t) for s in RANGER_SHAPES] for t in range(RANGER_STEPS)]
    grads[3][1][0, 0] = float("nan")
    grads[4][4][2, 5] = float("inf")
    grads[9][0][1, 1, 0] = float("-inf")
    return params, grads


def make_ranger_golden():
    """The reference's own Ranger class (lib/torch_utils/solver/ranger.py) stepped on CPU, two param groups."""
    ref_shim.install()
    from lib.torch_utils.solver.ranger import Ranger
    from lib.torch_utils.misc import nan_to_num

    params, grads = ranger_problem()
    ps = [torch.nn.Parameter(p.clone()) for p in params]
    opt = Ranger([dict(params=ps[:3], lr=2e-2), dict(params=ps[3:], lr=5e-3, weight_decay=0.1)], lr=1e-2)
    out = {}
    for t in range(RANGER_STEPS):
        for p, g in zip(ps, grads[t]):
            p.grad = g.clone()
            nan_to_num(p.grad, nan=0, posinf=1e5, neginf=-1e5, out=p.grad)  # engine.py:351-353
        opt.step()
        for i, p in enumerate(ps):
            out[f"p{i}_step{t + 1}"] = _np(p.data)
    for i, p in enumerate(ps):
        st = opt.state[p]
        out[f"exp_avg{i}"], out[f"exp_avg_sq{i}"], out[f"slow{i}"] = _np(st["exp_avg"]), _np(st["exp_avg_sq"]), _np(st["slow_buffer"])
    path = os.path.join(GOLDEN_DIR, "ranger_steps.npz")
    np.savez_compressed(path, **out)
    print(f"ranger_steps: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def make_rot_mats_golden(seed=41, B=24):
    """a10: the reference's ``get_rot_mat`` (models/model_utils.py:28-40) for every rotation type on seeded residuals
    (O(1) values, small-angle rows on both sides of the reference's thresholds, one exact zero), its autograd gradient
    for a fixed upstream G, and ``pose_scale_from_delta_init`` fed with those matrices (ego and allo)."""
    ref_shim.install()
    from core.catre.models.model_utils import get_rot_mat
    from core.catre.models.pose_scale_from_delta_init import pose_scale_from_delta_init

    g = torch.Generator().manual_seed(seed)
    arrays = {}
    G = torch.randn(B, 3, 3, generator=g)
    arrays["upstream"] = _np(G)
    R0 = None
    for name, d in (("rot6d", 6), ("quat", 4), ("log_quat", 3), ("lie_vec", 3)):
        r = torch.randn(B, d, generator=g)
        if d == 3:  # rows 0-5: small angles around the 1e-3 (theta^2 = 1e-6) switch of lie_vec / tiny log-quats
            r[0] *= 2e-4
            r[1] *= 5e-4
            r[2] *= 9e-4 / r[2].norm()
            r[3] *= 1.1e-3 / r[3].norm()
            r[4] *= 1e-6
            r[5] = 0.0
        if d == 4:
            r[0] *= 1e-3  # quat2mat_torch normalises: scale-invariant
            r[1] *= 50.0
        rr = r.clone().requires_grad_(True)
        R = get_rot_mat(rr, f"ego_{name}")
        (R * G).sum().backward()
        arrays[f"{name}_in"], arrays[f"{name}_R"], arrays[f"{name}_grad"] = _np(r), _np(R), _np(rr.grad)
        if R0 is None:
            R0 = R.detach()
        # the update with this residual, ego and allo (pose_scale_from_delta_init.py:87-93)
        t0 = torch.tensor([0.05, -0.03, 0.9]) + 0.05 * torch.randn(B, 3, generator=g)
        dt = torch.tensor([0.0, 0.0, 1.0]) + 0.02 * torch.randn(B, 3, generator=g)
        ds = 0.01 * torch.randn(B, 3, generator=g)
        s0 = 0.1 + 0.05 * torch.rand(B, 3, generator=g)
        K = torch.tensor([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]]).expand(B, 3, 3).contiguous()
        arrays[f"{name}_t0"], arrays[f"{name}_dt"], arrays[f"{name}_ds"], arrays[f"{name}_s0"] = map(_np, (t0, dt, ds, s0))
        for allo in (False, True):
            Rt, tt, st = pose_scale_from_delta_init(
                R.detach(), dt, ds, R0, t0, s0, Ks=K, K_aware=True, delta_T_space="image", delta_T_weight=1.0,
                delta_z_style="cosypose", eps=1e-4, is_allo=allo, scale_type="iter_add")
            tag = f"{name}_{'allo' if allo else 'ego'}"
            arrays[f"{tag}_R"], arrays[f"{tag}_t"], arrays[f"{tag}_s"] = _np(Rt), _np(tt), _np(st)
    arrays["R0"], arrays["K"] = _np(R0), _np(K)
    path = os.path.join(GOLDEN_DIR, "rot_mats.npz")
    np.savez_compressed(path, **arrays)
    print(f"rot_mats: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); NaN grads: "
          + ", ".join(f"{k}={int(np.isnan(v).sum())}" for k, v in arrays.items() if k.endswith("_grad")))


def bottle_prior():
    """The reference's data file for category 'bottle' (config 1 of BASELINE.json)."""
    import pickle

    p = os.path.join(ref_shim.REFERENCE_ROOT, "datasets/NOCS/obj_models/cr_normed_mean_model_points_spd.pkl")
    with open(p, "rb") as f:
        return np.asarray(pickle.load(f)["bottle"], dtype=np.float32)


def main(argv=None):
    names = (argv or sys.argv[1:]) or list(CASES)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(8)
    if "ranger" in names:
        make_ranger_golden()
        names = [n for n in names if n != "ranger"]
    if "pcl" in names:
        make_pcl_golden()
        names = [n for n in names if n != "pcl"]
    if "aug" in names:
        make_aug_golden()
        names = [n for n in names if n != "aug"]
    if "rot_mats" in names:
        make_rot_mats_golden()
        names = [n for n in names if n != "rot_mats"]
    if "amp" in names:
        make_amp_golden()
        names = [n for n in names if n != "amp"]
    if "amp_train" in names:
        for n in TRAIN_CASES:
            make_amp_train_golden(n)
        names = [n for n in names if n != "amp_train"]
    if not (argv or sys.argv[1:]):
        names = names + list(TRAIN_CASES)
        make_ranger_golden()
        make_amp_golden()
        for n in TRAIN_CASES:
            make_amp_train_golden(n)
        make_aug_golden()
        make_pcl_golden()
        make_rot_mats_golden()
    for name in names:
        if name in TRAIN_CASES:
            make_train_golden(name)
            continue
        B, N, M, K, seed, salt, prior, ov = CASES[name]
        pr = bottle_prior() if prior == "bottle" else None
        batch = synth.make_inputs(B, N, M, seed=seed, prior=pr)
        cfg = reference_cfg(N, M, ov)
        out = run_reference(cfg, batch, K, salt)
        arrays = {f"in_{k}": _np(v) for k, v in batch.items()}
        arrays.update(out)
        arrays["meta_overrides"] = np.array(repr(sorted(ov.items())))
        arrays["meta"] = np.array([B, N, M, K, seed, salt], dtype=np.int64)
        path = os.path.join(GOLDEN_DIR, f"{name}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
        print("   pose_K[0] =", np.array2string(out[f"pose_{K}"][0], precision=4).replace("\n", " "))
        print("   scale_K[0]=", out[f"scale_{K}"][0], " rot6d=", out["stage_rot_deltas"][0],
              " dt=", out["stage_trans_deltas"][0], " ds=", out["stage_scale_deltas"][0])


if __name__ == "__main__":
    main()
