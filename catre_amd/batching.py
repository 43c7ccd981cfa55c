"""Loop glue either side of the model - counterparts of reference ``core/catre/engine/batch_test.py:63-99``
(and the identical train-side math, ``engine/batching.py:128-144``)."""
import torch

from .runtime import pose_apply


def batch_updater_test(cfg, batch, poses_est=None, scales_est=None, device="cuda", dtype=torch.float32):
    """In-place update of ``batch`` for the next refine iteration: feeds back the estimates and
    writes ``batch["tfd_kps"] [B,3,M]`` and ``batch["x"] [B,3,N]`` (permuted views, like the reference)."""
    if poses_est is not None:
        batch["obj_pose_est"] = poses_est
    if scales_est is not None and cfg.MODEL.REFINE_SCLAE:
        batch["obj_scale_est"] = scales_est
    if "obj_kps" not in batch:
        raise KeyError("batch['obj_kps'] missing (the reference fills it with get_normed_kps, engine_utils.py:17-35)")
    x, tfd_kps = pose_apply(batch["pcl"], batch["obj_kps"], batch["obj_pose_est"], batch["obj_scale_est"],
                            zero_center=cfg.INPUT.ZERO_CENTER_INPUT)
    batch["tfd_kps"] = tfd_kps
    batch["x"] = x


batch_updater = batch_updater_test
