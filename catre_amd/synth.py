"""Seeded synthetic weights and point clouds for tests, goldens and ``bench.py``.

Nothing here comes from the reference: the pretrained checkpoint and the NOCS frames
are absent from the reference tree (``/root/reference/.MISSING_LARGE_BLOBS``), so parity
and throughput are measured on procedurally generated inputs of the reference's shapes
(SURVEY.md section 8c "Golden-vector design").

* :func:`recipe_state_dict` - "trained-scale" weights: every tensor is drawn from a CPU
  ``torch.Generator`` seeded with ``crc32(key)``, scaled so that the residual heads emit
  deltas of a few degrees / centimetres (fresh reference init would predict ``t = 0``,
  see SURVEY.md section 8b "Init").
* :func:`make_inputs` - posed shape prior + noise as the observed cloud, perturbed
  initial pose/scale, NOCS intrinsics (values from reference ``ref/nocs.py:103``).
"""
import math
import zlib

import torch

NOCS_K = ((591.0125, 0.0, 322.525), (0.0, 590.16775, 244.11084), (0.0, 0.0, 1.0))


def _gen(key, salt=0):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


def recipe_tensor(key, shape, salt=0):
    """One parameter tensor of the weight recipe (fp32, CPU)."""
    shape = tuple(shape)
    g = _gen(key, salt)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    leaf = key.rsplit(".", 1)[-1]
    fan_in = 1
    if leaf == "weight" and len(shape) >= 2:
        fan_in = int(torch.tensor(shape[1:]).prod())

    if ".norm." in key or key.endswith("norm.weight") or key.endswith("norm.bias"):
        # the never-used GroupNorm of each head (reference conv_out_per_rot_head.py:92)
        return torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
    if key.startswith("pcl_net."):
        if leaf == "weight":
            if ".fc3." in key:  # STN output layer: T = I + small
                return r * (0.25 / math.sqrt(fan_in))
            return r * math.sqrt(2.0 / fan_in)
        if ".fc3." in key:
            return r * 0.02
        return r * 0.05
    if "conv_p" in key:
        if leaf == "weight":
            return (1.0 + 0.5 * r) / shape[1]
        return r * 0.01
    if ".neck." in key:
        if leaf == "weight":
            return r * (0.3 / math.sqrt(fan_in))
        base = torch.zeros(shape)
        base[0 if "rot_head_x" in key else 1] = 1.0
        return base + 0.02 * r
    if ".fc_t." in key:
        if leaf == "weight":
            return r * (0.05 / math.sqrt(fan_in))
        return torch.tensor([0.0, 0.0, 1.0]).repeat(shape[0] // 3) + 0.002 * r
    if ".fc_s." in key:
        if leaf == "weight":
            return r * (0.02 / math.sqrt(fan_in))
        return 0.001 * r
    # head conv / linear layers and their GroupNorm affine
    if leaf == "weight":
        if len(shape) == 1:  # GroupNorm gamma
            return 1.0 + 0.1 * r
        return r / math.sqrt(fan_in)
    if len(shape) == 1 and (".layers.1." in key or ".layers.4." in key or ".linears.1." in key or ".linears.4." in key):
        return 0.1 * r  # GroupNorm beta
    return 0.05 * r


def recipe_state_dict(shapes, salt=0):
    """``shapes``: mapping key -> shape (e.g. ``{k: v.shape for k, v in model.state_dict().items()}``)."""
    return {k: recipe_tensor(k, tuple(s), salt) for k, s in sorted(shapes.items())}


def _quat_to_mat(q):
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    ).reshape(q.shape[:-1] + (3, 3))


def _axis_angle(v):
    """Rodrigues, v:[B,3] (angle = |v|)."""
    th = v.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    k = v / th
    K = torch.zeros(v.shape[0], 3, 3, dtype=v.dtype)
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    th = th.unsqueeze(-1)
    return torch.eye(3, dtype=v.dtype) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)


def procedural_prior(M, gen, kind="cylinder"):
    """A normalised shape prior in [-0.5, 0.5]^3, [M,3] fp32 (stand-in for the reference's
    ``cr_normed_mean_model_points_spd.pkl`` entries)."""
    u = torch.rand(M, generator=gen)
    v = torch.rand(M, generator=gen)
    if kind == "cylinder":  # bottle / can like: lateral surface + caps
        ang = 2 * math.pi * u
        cap = v > 0.8
        rad = torch.where(cap, 0.2 * torch.sqrt(torch.rand(M, generator=gen)), torch.full((M,), 0.2))
        y = torch.where(cap, torch.where(u > 0.5, torch.tensor(0.5), torch.tensor(-0.5)), v / 0.8 - 0.5)
        pts = torch.stack([rad * torch.cos(ang), y, rad * torch.sin(ang)], -1)
    else:  # box surface (laptop / camera like)
        p = torch.rand(M, 3, generator=gen) - 0.5
        face = torch.randint(0, 3, (M,), generator=gen)
        sign = torch.where(torch.rand(M, generator=gen) > 0.5, 0.5, -0.5)
        p[torch.arange(M), face] = sign
        pts = p * torch.tensor([1.0, 0.6, 0.9])
    return pts.float()


def make_inputs(B, N=1024, M=1024, seed=0, prior=None, dtype=torch.float32):
    """Synthetic refine-loop inputs with the reference's batch keys (``engine/batch_test.py:10-60``).

    Returns a dict of CPU tensors: ``pcl [B,N,3]``, ``obj_kps [B,M,3]``, ``obj_pose_est [B,3,4]``,
    ``obj_scale_est [B,3]``, ``K [B,3,3]``, ``obj_mean_scales [B,3]``, ``obj_cls [B]``,
    plus the ground truth used to pose the cloud (``gt_rot``, ``gt_trans``, ``gt_scale``).
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(1000003 * seed + 17)
    if prior is None:
        kps = torch.stack(
            [procedural_prior(M, g, "cylinder" if (b % 2 == 0) else "box") for b in range(B)], 0
        )
    else:
        prior = torch.as_tensor(prior, dtype=torch.float32)
        assert prior.shape == (M, 3), prior.shape
        kps = prior.unsqueeze(0).repeat(B, 1, 1)
    gt_rot = _quat_to_mat(torch.randn(B, 4, generator=g))
    gt_trans = torch.tensor([0.0, 0.0, 1.0]) + 0.1 * torch.randn(B, 3, generator=g)
    mean_scales = torch.tensor([0.087, 0.220, 0.089]).repeat(B, 1)
    gt_scale = mean_scales * (1.0 + 0.15 * torch.randn(B, 3, generator=g)).clamp(0.6, 1.4)

    # observed cloud: N points re-sampled (with replacement) from the posed prior + 2 mm noise
    idx = torch.randint(0, M, (B, N), generator=g)
    src = torch.gather(kps, 1, idx.unsqueeze(-1).expand(B, N, 3))
    pcl = (gt_rot.unsqueeze(1) @ (src * gt_scale.unsqueeze(1)).unsqueeze(-1)).squeeze(-1)
    pcl = pcl + gt_trans.unsqueeze(1) + 0.002 * torch.randn(B, N, 3, generator=g)

    # initial estimate = ground truth perturbed (~10 deg, 1 cm, 5 mm)
    dR = _axis_angle(torch.randn(B, 3, generator=g) * math.radians(10.0) / math.sqrt(3.0))
    rot0 = dR @ gt_rot
    t0 = gt_trans + 0.01 * torch.randn(B, 3, generator=g)
    s0 = (gt_scale + 0.005 * torch.randn(B, 3, generator=g)).clamp_min(0.04)
    pose0 = torch.cat([rot0, t0.unsqueeze(-1)], -1)

    out = dict(
        pcl=pcl,
        obj_kps=kps,
        obj_pose_est=pose0,
        obj_scale_est=s0,
        K=torch.tensor(NOCS_K).repeat(B, 1, 1),
        obj_mean_scales=mean_scales,
        gt_rot=gt_rot,
        gt_trans=gt_trans,
        gt_scale=gt_scale,
    )
    out = {k: v.to(dtype).contiguous() for k, v in out.items()}
    out["obj_cls"] = torch.arange(B, dtype=torch.long) % 6
    return out


def y_axis_symmetries(n):
    """n-1 rotations about the y axis by multiples of 2 pi / n, float32 [n-1,3,3]: the symmetry transformations of the NOCS
    y-symmetric categories (bottle / bowl / can, ref/nocs.py:138-158; shape of lib/pysixd/misc.py:220-231) as synthetic
    `sym_info` entries for benches and probes."""
    import numpy as np

    a = 2.0 * np.pi * np.arange(1, n, dtype=np.float64) / n
    out = np.zeros((n - 1, 3, 3), dtype=np.float32)
    out[:, 0, 0], out[:, 0, 2], out[:, 1, 1], out[:, 2, 0], out[:, 2, 2] = np.cos(a), np.sin(a), 1.0, -np.sin(a), np.cos(a)
    return out


def make_depth_scene(H=120, W=160, n_inst=5, seed=0):
    """A synthetic depth frame for the point-cloud preparation: a tilted background plane with
    ``n_inst`` ellipsoidal blobs in front of it, zero-depth holes, per-instance masks and poses.  The instances cover
    the branches of the reference's ball crop: 0 regular; 1 a tiny object (radius floor 0.05 + growth); 2 a pose centre
    a little off its mask (radius growth); 3 a pose centre far away (falls back to every masked pixel); 4 very few
    masked pixels (candidate list tiled up to the sample count)."""
    g = torch.Generator().manual_seed(seed)
    K = torch.tensor([[0.9 * W, 0.0, W / 2 - 0.5], [0.0, 0.9 * W, H / 2 - 0.5], [0.0, 0.0, 1.0]], dtype=torch.float32)
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    depth = 1.6 + 0.002 * (u - W / 2) + 0.001 * (v - H / 2)
    masks = torch.zeros(n_inst, H, W, dtype=torch.bool)
    poses = torch.zeros(n_inst, 3, 4)
    scales = torch.zeros(n_inst, 3)
    for i in range(n_inst):
        cu, cv = W * (0.15 + 0.7 * (i + 0.5) / n_inst), H * (0.3 + 0.4 * torch.rand(1, generator=g).item())
        ru, rv = W * 0.07, H * 0.12
        if i % 5 == 4:
            ru, rv = 2.2, 1.6
        inside = ((u - cu) / ru) ** 2 + ((v - cv) / rv) ** 2 < 1
        zc = 0.9 + 0.1 * i
        bump = zc - 0.05 * torch.sqrt(torch.clamp(1 - ((u - cu) / ru) ** 2 - ((v - cv) / rv) ** 2, min=0))
        depth = torch.where(inside, bump, depth)
        masks[i] = inside
        q = torch.randn(4, generator=g)
        q = q / q.norm()
        w, x, y, z = q.tolist()
        R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        t = torch.tensor([(cu - K[0, 2]) * zc / K[0, 0], (cv - K[1, 2]) * zc / K[1, 1], zc])
        s = torch.tensor([0.12, 0.2, 0.15])
        if i % 5 == 1:
            s = s * 0.05
        if i % 5 == 2:
            t = t + torch.tensor([0.16, 0.0, 0.0])
        if i % 5 == 3:
            t = t + torch.tensor([0.0, 0.9, 0.5])
        poses[i, :, :3], poses[i, :, 3], scales[i] = R, t, s
    holes = torch.rand(H, W, generator=g) < 0.03
    depth = torch.where(holes, torch.zeros_like(depth), depth).to(torch.float32)
    return dict(depth=depth, K=K, masks=masks, poses=poses, scales=scales)
