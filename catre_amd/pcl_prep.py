"""Point-cloud preparation on the device - what the reference's data loader does per instance on
the CPU (``core/catre/datasets/data_loader.py:576-603``): back-project the depth map, keep the instance's masked pixels
with depth > 0, crop a ball around the pose centre (radius grown x1.1 until it holds >= 10 points) and sample
``NUM_PCL`` of them (``core/utils/cat_data_utils.py:209-226,289-320,352-400``; ``lib/pysixd/misc.py:360-378``).

All instances of a frame go through four launches.  ``sample="host"`` draws ``torch.randperm`` per instance exactly
like the reference (same global-generator consumption, bit-identical clouds; costs one device->host copy of the
candidate counts); ``sample="device"`` uses a keyed permutation evaluated on the GPU (no host round trip, a different
but equally uniform sample without replacement).
"""
import ctypes

import torch

from . import hip


def backproject_th(depth, K):
    """``lib/pysixd/misc.py:360-378``: organised cloud map [H,W,3] (plain tensor ops; not on the hot path)."""
    assert depth.ndim == 2, depth.ndim
    H, W = depth.shape
    Y, X = torch.meshgrid(torch.arange(H, device=depth.device, dtype=depth.dtype) - float(K[1][2]),
                          torch.arange(W, device=depth.device, dtype=depth.dtype) - float(K[0][2]), indexing="ij")
    return torch.stack((X * depth / float(K[0][0]), Y * depth / float(K[1][1]), depth), dim=2)


def _k9(K):
    K = torch.as_tensor(K, dtype=torch.float32).reshape(3, 3).cpu()
    return (ctypes.c_float * 9)(*[float(v) for v in K.reshape(-1)])


def _random_sample(n, npoint):
    """``random_sample`` of the reference (``cat_data_utils.py:322-329``), same random stream."""
    idx = torch.randperm(n)[:npoint]
    while len(idx) < npoint:
        idx = torch.cat((idx, _random_sample(n, npoint - len(idx))), dim=0)
    return idx


def sample_instances(depth, K, masks, poses=None, scales=None, ratio=0.5, num_points=1024, use_ball=True, sample="host",
                     seed=0, fps_sample=False, return_pixels=False):
    """depth [H,W] (device fp32, metres), K 3x3, masks [I,H,W] bool/uint8 (or None: whole frame, I from poses),
    poses [I,3,4], scales [I,3] -> pcl [I,num_points,3] (+ flat pixel indices [I,num_points] with ``return_pixels``).
    ``use_ball=True`` = ``crop_ball_from_depth_image`` (``INPUT.SAMPLE_DEPTH_FROM_BALL``), ``False`` =
    ``crop_mask_depth_image``."""
    if sample not in ("host", "device"):
        raise ValueError(f"sample={sample!r}: expected 'host' or 'device'")
    lib = hip.load()
    depth = hip.require_dev_f32(depth.contiguous(), "depth")
    H, W = depth.shape
    dev = depth.device
    if masks is not None:
        if masks.dtype == torch.bool:
            masks = masks.to(torch.uint8)
        if masks.dtype != torch.uint8 or masks.device != dev or masks.shape[1:] != (H, W):
            raise ValueError("masks must be [I,H,W] bool / uint8 on the depth map's device")
        masks = masks.contiguous()
        I = masks.shape[0]
    else:
        I = poses.shape[0]
    if poses is None or scales is None:
        if use_ball:
            raise ValueError("the ball crop needs poses and scales")
        poses = torch.zeros(I, 3, 4, device=dev)
        scales = torch.ones(I, 3, device=dev)
    poses = hip.require_dev_f32(poses.contiguous(), "poses", (I, 3, 4))
    scales = hip.require_dev_f32(scales.contiguous(), "scales", (I, 3))
    k9 = _k9(K)
    nbytes = lib.catre_pcl_workspace_bytes(I, H, W)
    ws = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    counts = torch.empty(I, dtype=torch.int32, device=dev)
    st = hip.stream_ptr(dev)
    hip.check(lib.catre_pcl_candidates(hip.ptr(depth), k9, hip.ptr(masks), hip.ptr(poses), hip.ptr(scales), float(ratio),
                                       int(bool(use_ball)), I, H, W, hip.ptr(ws), nbytes, hip.ptr(counts), st),
              "catre_pcl_candidates")
    sidx = None
    if fps_sample:
        # INPUT.FPS_SAMPLE: farthest point sampling of the tiled candidate list (cat_data_utils.py:305-306 with
        # device="cpu" as the data loader passes -> farthest_points_torch.py:6-62), one workgroup per instance
        cl = counts.cpu().tolist()
        if min(cl) == 0:
            raise ValueError("an instance has no masked pixel with depth > 0")
        cap = 0
        for c in cl:
            L = c
            while L < num_points:
                L *= 2
            cap = max(cap, L)
        scratch = torch.empty(I * 4 * cap, dtype=torch.float32, device=dev)
        sidx = torch.empty(I, num_points, dtype=torch.int64, device=dev)
        hip.check(lib.catre_pcl_fps(hip.ptr(depth), k9, hip.ptr(ws), nbytes, I, H, W, num_points, hip.ptr(scratch), cap,
                                    hip.ptr(sidx), st), "catre_pcl_fps")
    elif sample == "host":
        rows = []
        for c in counts.cpu().tolist():  # instance order = the data loader's loop order
            if c == 0:
                # the reference recurses with a 1.2x larger ratio for ever here (cat_data_utils.py:390-393)
                raise ValueError("an instance has no masked pixel with depth > 0")
            if use_ball:  # the candidate list is tiled until it holds num_points entries, then one permutation
                L = c
                while L < num_points:
                    L *= 2
                rows.append(torch.randperm(L)[:num_points])  # cat_data_utils.py:301-309, 322-329
            else:        # crop_mask_depth_image: random_sample tops a short list up with further permutations
                rows.append(_random_sample(c, num_points))
        sidx = torch.stack(rows).to(dev)
    pcl = torch.empty(I, num_points, 3, dtype=torch.float32, device=dev)
    pix = torch.empty(I, num_points, dtype=torch.int32, device=dev) if return_pixels else None
    hip.check(lib.catre_pcl_sample(hip.ptr(depth), k9, hip.ptr(ws), nbytes, hip.ptr(sidx), int(seed) & (2**64 - 1), I, H, W,
                                   num_points, hip.ptr(pcl), hip.ptr(pix), st), "catre_pcl_sample")
    if return_pixels:
        return pcl, pix, counts
    return pcl


def crop_ball_from_depth_image(image, depth, mask, pose, scale, ratio, cam_intrinsics, coord=None, num_points=None,
                               device=None, fps_sample=False):
    """Single-instance signature of the reference (``cat_data_utils.py:380-400``) on top of :func:`sample_instances`;
    ``depth`` is the [H,W] depth map or the [H,W,3] cloud map the reference passes.  -> (rgb, pts, nocs)."""
    d = depth[..., 2] if depth.ndim == 3 else depth
    pcl, pix, _ = sample_instances(d, cam_intrinsics, mask[None], pose[None], scale[None], ratio=ratio,
                                   num_points=num_points, use_ball=True, sample="host", fps_sample=fps_sample,
                                   return_pixels=True)
    flat = pix[0].long()
    rgb = image.reshape(-1, image.shape[-1])[flat.to(image.device)] if image is not None else None
    nocs = coord.reshape(-1, 3)[flat.to(coord.device)] if coord is not None else None
    return rgb, pcl[0], nocs
