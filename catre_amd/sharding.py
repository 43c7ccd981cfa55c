"""Multi-GPU sharding of the refine path: one process per GPU, objects split across ranks.

Every object (batch row) is independent through the whole path (no BatchNorm; GroupNorm is per
sample - SURVEY.md section 8e), so inference shards with NO data-path collective: each rank refines
its contiguous slice of the global batch.  The only exchange is the optional all-gather of the
results at the end, the counterpart of the reference's prediction all-gather
(``core/catre/engine/catre_custom_evaluator.py:203``).  ``torch.distributed`` backend ``"nccl"`` is
RCCL over xGMI on MI355X; ``"gloo"`` is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous [lo, hi) slice of ``total`` objects owned by ``rank``; sizes differ by at most one
    (the reference hands each rank ``IMS_PER_BATCH / world`` images, ``core/utils/dataset_utils.py:411-416``)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# per-object entries of the reference's batch dicts (engine/batch_test.py:10-60,63-99; engine/batching.py:9-146): first
# dimension = the object.  Per-IMAGE entries ("img", "depth_obs", "roi_img", ...) and scalars are never sliced, whatever
# their length happens to be.
PER_OBJECT_KEYS = frozenset((
    "obj_cls", "obj_bbox", "obj_pose", "obj_scale", "obj_pose_est", "obj_scale_est", "obj_mean_points", "obj_mean_scales",
    "obj_fps_points", "obj_kps", "im_id", "inst_id", "K", "sym_info", "pcl", "x", "tfd_kps",
    "gt_rot", "gt_trans", "gt_scale", "obj_pose_gt", "obj_scale_gt", "nocs_scale",
    "last_frame_poses", "obj_visib_mask", "obj_trunc_mask",       # batching.py:29-41 (concatenated over instances)
))
# per-IMAGE entries (batching.py:12-27): never sliced, and never warned about when an image count happens to equal B
# (plus the per-image meta entries of the reference's dataset dicts that evaluator-side code carries along:
# core/catre/datasets/data_loader.py dataset_dict keys)
PER_IMAGE_KEYS = frozenset(("img", "depth_obs", "roi_img", "roi_depth", "file_name", "scene_im_id", "cam", "im_H", "im_W",
                            "depth_file", "depth_factor", "img_type", "dataset_name", "image_id", "width", "height",
                            "time", "resize_ratio"))


def shard_batch(batch, rank, world, extra_keys=(), per_image_keys=()):
    """Slice the per-object tensors / lists of a reference-style batch dict (``PER_OBJECT_KEYS`` + ``extra_keys``);
    per-image entries (``PER_IMAGE_KEYS`` + ``per_image_keys``), scalars and anything whose length is not the object
    count pass through untouched; an unlisted entry with exactly B rows raises (ambiguous)."""
    B = batch["pcl"].shape[0]
    lo, hi = shard_bounds(B, rank, world)
    keys = PER_OBJECT_KEYS | frozenset(extra_keys)
    out = {}
    for k, v in batch.items():
        if k not in keys:
            # an unknown entry whose first dimension is the object count is most likely a per-object entry the caller
            # added: passing it through whole would pair B_local objects with B_global rows - refuse instead of guessing
            n = v.shape[0] if isinstance(v, torch.Tensor) and v.dim() > 0 else (len(v) if isinstance(v, (list, tuple)) else -1)
            if n == B and world > 1 and k not in PER_IMAGE_KEYS and k not in per_image_keys:
                raise ValueError(f"batch[{k!r}] has {B} rows like a per-object entry but is not in PER_OBJECT_KEYS: pass "
                                 f"extra_keys=({k!r},) to slice it or per_image_keys=({k!r},) to keep it whole")
            out[k] = v
            continue
        n = v.shape[0] if isinstance(v, torch.Tensor) else len(v)
        if n != B:
            raise ValueError(f"batch[{k!r}] is a per-object entry but holds {n} rows for {B} objects")
        out[k] = v[lo:hi].contiguous() if isinstance(v, torch.Tensor) else v[lo:hi]
    return out


def gather_outputs(local_out, total, group=None):
    """All-gather an ``out_dict`` (``pose_i`` / ``scale_i`` tensors with the local objects first) so that
    every rank holds the full-batch result in the original object order.  Ragged shards are padded to
    the largest shard for the collective."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    full = {}
    for k, v in local_out.items():
        pad = torch.zeros((maxn,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[: v.shape[0]] = v
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        full[k] = torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)
    return full


def refine_sharded(refine_fn, batch, n_iter, group=None, gather=True):
    """Run ``refine_fn(local_batch, n_iter) -> out_dict`` on this rank's shard of ``batch``."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    local = refine_fn(shard_batch(batch, rank, world), n_iter)
    if not gather:
        return local
    return gather_outputs(local, batch["pcl"].shape[0], group)
