"""ORACLE (test infrastructure only) - restatement of the reference's per-instance point-cloud preparation:
``backproject_th`` (``lib/pysixd/misc.py:360-378``), ``sample_bp_depth`` (``core/utils/cat_data_utils.py:209-226``),
``crop_ball_from_pts`` (``:289-320``), ``crop_ball_from_depth_image`` (``:380-400``) and ``crop_mask_depth_image``
(``:352-377``), with the ``torch.randperm`` draw made explicit, and the farthest point sampling the data loader uses
with ``INPUT.FPS_SAMPLE`` (``core/utils/farthest_points_torch.py:6-62``).  Pinned by ``tests/golden/pcl_prep.npz``."""
import torch


def backproject(depth, K):
    H, W = depth.shape
    Y, X = torch.meshgrid(torch.arange(H, dtype=depth.dtype) - K[1, 2], torch.arange(W, dtype=depth.dtype) - K[0, 2],
                          indexing="ij")
    return torch.stack((X * depth / K[0, 0], Y * depth / K[1, 1], depth), dim=2)


def candidates(depth, K, mask, pose=None, scale=None, ratio=0.5, use_ball=True):
    """-> flat pixel indices (row-major order) the reference would sample from, and the [H*W,3] cloud map."""
    bp = backproject(depth, K).reshape(-1, 3)
    valid = bp[:, 2] > 0
    if mask is not None:
        valid = torch.logical_and(mask.reshape(-1), valid)
    pix = valid.nonzero().reshape(-1)
    if not use_ball:
        return pix, bp
    pts = bp[pix]
    centre = pose[:, 3]
    radius = ratio * torch.norm(pose[:, :3] @ scale)
    distance = torch.sqrt(((pts - centre) ** 2).sum(-1))
    radius = max(radius, 0.05)
    for _ in range(10):
        idx = torch.where(distance <= radius)[0]
        if len(idx) >= 10:
            break
        radius *= 1.10
    if len(idx) == 0:
        idx = torch.where(distance <= 1e9)[0]
    return pix[idx], bp


def tiled_length(count, num_points):
    L = count
    while 0 < L < num_points:
        L *= 2
    return L


def random_sample_idx(n, npoint):
    """``random_sample`` (``cat_data_utils.py:322-329``): a random permutation cut to npoint; a list shorter than
    npoint is topped up by further (recursive) draws - each consuming one ``torch.randperm(n)``."""
    idx = torch.randperm(n)[:npoint]
    while len(idx) < npoint:
        idx = torch.cat((idx, random_sample_idx(n, npoint - len(idx))), dim=0)
    return idx


def sample(pix, bp, sample_idx):
    """``idx`` doubled until it holds num_points entries, then ``idx[sample_idx]`` (:309-319)."""
    if len(pix) == 0:
        return torch.zeros(len(sample_idx), 3, dtype=bp.dtype), torch.full((len(sample_idx),), -1, dtype=torch.long)
    sel = pix[sample_idx % len(pix)]
    return bp[sel], sel


def farthest_points(points, n):
    """``farthest_points(data, n_clusters=n, dist_func=F.pairwise_distance, init_center=True)`` of
    ``core/utils/farthest_points_torch.py:6-62`` as ``crop_ball_from_pts`` calls it with ``device="cpu"``
    (``cat_data_utils.py:305-306, 331-341``): -> the n centre indices in the order they are picked."""
    import torch.nn.functional as F

    if n >= points.shape[0]:
        return torch.arange(points.shape[0], dtype=torch.long)
    dist = F.pairwise_distance(points.mean(0, keepdim=True).expand(points.shape[0], -1), points)
    picks = torch.zeros(n, dtype=torch.long)
    for i in range(n):
        c = torch.argmax(dist)
        picks[i] = c
        dist = torch.min(dist, F.pairwise_distance(points[c].unsqueeze(0).expand(points.shape[0], -1), points))
    return picks


def fps_sample_idx(pix, bp, num_points):
    """slot indices into the tiled candidate list that ``crop_ball_from_pts(..., fps_sample=True)`` selects."""
    L = tiled_length(len(pix), num_points)
    slots = torch.arange(L) % len(pix)
    return farthest_points(bp[pix[slots]], num_points)
