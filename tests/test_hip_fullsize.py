"""Full-size properties of the TRAINING path (BASELINE.json config 3: B=256, N=M=1024, DDP world 1) and of buffers
past 2^31 elements in the backward - the shapes no oracle comparison can reach in seconds, checked through
size-independent properties: objects are independent of their batch (also in the gradients), runs are bitwise
deterministic (no atomics anywhere), everything is finite."""
import numpy as np
import pytest
import torch

from tests.util import recipe_sd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(N=1024, M=1024, compute=None):
    from catre_amd.CATRE_disR_shared import build_model_optimizer
    from catre_amd.config import default_cfg

    cfg = default_cfg(num_pcl=N, num_kps=M, device=DEV)
    if compute:
        cfg.MODEL.CATRE.COMPUTE_DTYPE = compute
    model, opt = build_model_optimizer(cfg, is_test=False)
    model.load_state_dict({k: v.to(DEV) for k, v in recipe_sd(cfg, 0).items()}, strict=True)
    return model.train(), opt, cfg


def _sym_info(B):
    from oracle.catre_oracle import y_axis_symmetries

    sym = y_axis_symmetries(314)
    return [sym if (i % 6) in (0, 1, 3) else None for i in range(B)]  # bottle / bowl / can (ref/nocs.py:138-158)


def _train_iter(model, cfg, b, sym, it=1):
    from catre_amd.batching import batch_updater_test

    b = dict(b)
    batch_updater_test(cfg, b)
    out, ld = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                    gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
                    mean_scales=b["obj_mean_scales"], sym_info=sym, do_loss=True, cur_iter=it)
    model.zero_grad(set_to_none=True)
    sum(ld.values()).backward()
    return out, ld, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


def _subset_grads(model, cfg, b, idx, G):
    """d/d params of sum(pose[idx] * G) - the forward in grad mode without the loss (its mean couples all objects)."""
    from catre_amd.batching import batch_updater_test

    b = dict(b)
    batch_updater_test(cfg, b)
    out = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                mean_scales=b["obj_mean_scales"], do_loss=False, cur_iter=1)
    model.zero_grad(set_to_none=True)
    ((out["pose_1"][idx] * G[0]).sum() + (out["scale_1"][idx] * G[1]).sum()).backward()
    return out, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("compute", [None, "split", "bf16"])
def test_config3_training_step_at_full_size(compute):
    """B=256, N=M=1024, do_loss=True (313 symmetry candidates for half the objects): finite, bitwise deterministic, and an
    optimizer step on it changes the weights."""
    from catre_amd import synth

    model, opt, cfg = _model(compute=compute)
    B = 256
    b = {k: v.to(DEV) for k, v in synth.make_inputs(B, 1024, 1024, seed=77).items()}
    sym = _sym_info(B)
    out1, ld1, g1 = _train_iter(model, cfg, b, sym)
    out2, ld2, g2 = _train_iter(model, cfg, b, sym)
    assert set(ld1) == {"loss_PM_R", "loss_rot", "loss_yaxis_rot", "loss_trans_xy", "loss_trans_z", "loss_scale"}
    assert all(torch.isfinite(v) for v in ld1.values()) and torch.isfinite(out1["pose_1"]).all()
    assert len(g1) == 68 and all(torch.isfinite(v).all() for v in g1.values())
    for k in ld1:
        assert torch.equal(ld1[k], ld2[k]), k
    for k in g1:
        assert torch.equal(g1[k], g2[k]), f"gradient of {k} is not reproducible run to run"
    R = out1["pose_1"][:, :3, :3]
    assert (R @ R.transpose(1, 2) - torch.eye(3, device=DEV)).abs().max() < 1e-4
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    changed = sum(int(not torch.equal(before[k], p)) for k, p in model.named_parameters())
    assert changed == 68 and all(torch.isfinite(p).all() for p in model.parameters())


def test_config3_gradients_of_a_subset_do_not_depend_on_the_batch():
    """Gradients that 5 objects send to the parameters inside the B=256 batch == the same 5 objects run alone
    (objects are independent: no BatchNorm, per-sample GroupNorm)."""
    from catre_amd import synth

    model, _, cfg = _model()
    B = 256
    b = {k: v.to(DEV) for k, v in synth.make_inputs(B, 1024, 1024, seed=78).items()}
    idx = torch.tensor([0, 17, 128, 254, 255], device=DEV)
    g = torch.Generator().manual_seed(2)
    G = (torch.randn(5, 3, 4, generator=g).to(DEV), torch.randn(5, 3, generator=g).to(DEV))
    out_b, grads_b = _subset_grads(model, cfg, b, idx, G)
    sub = {k: v[idx].contiguous() for k, v in b.items()}
    out_s, grads_s = _subset_grads(model, cfg, sub, torch.arange(5, device=DEV), G)
    assert (out_b["pose_1"][idx] - out_s["pose_1"]).abs().max() <= 1e-6
    assert set(grads_b) == set(grads_s)
    for k in grads_s:
        ref = grads_s[k]
        err = float((grads_b[k] - ref).abs().max()) / (float(ref.abs().max()) + 1e-20)
        # the same 10 k non-zero rows, but grouped differently over the wgrad workgroups (512 k rows in the batch): fp32
        # re-association only.  First-layer gradients are sums with heavy cancellation (measured 2.8e-4 of the max entry)
        # (up to 5e-3 of the max entry on the [64,3] conv1 weights, 1e-4 in Frobenius norm): the bound separates "re-association" from "coupled objects", where whole
        # rows of other objects would leak in at O(1)
        assert err <= 2e-2, (k, err)
        nrm = float((grads_b[k] - ref).norm()) / (float(ref.norm()) + 1e-20)
        assert nrm <= 2e-3, (k, nrm)


def test_training_buffers_past_2_to_the_31_elements():
    """B=2100 at N=M=1024: the saved [rows, 512] activation holds 4.3 M x 512 = 2.2e9 floats (> 2^31), the iteration
    ~58 GB of the 288 GB.  The LAST objects of the batch (offsets past 2^31 in forward and backward) must behave like
    the same objects run alone."""
    from catre_amd import synth

    free, _ = torch.cuda.mem_get_info()
    if free < 90e9:
        pytest.skip("needs ~60 GB of free HBM")
    model, _, cfg = _model()
    B = 2100
    b = {k: v.to(DEV) for k, v in synth.make_inputs(B, 1024, 1024, seed=79).items()}
    idx = torch.tensor([B - 3, B - 2, B - 1], device=DEV)
    g = torch.Generator().manual_seed(4)
    G = (torch.randn(3, 3, 4, generator=g).to(DEV), torch.randn(3, 3, generator=g).to(DEV))
    out_b, grads_b = _subset_grads(model, cfg, b, idx, G)
    assert torch.isfinite(out_b["pose_1"]).all()
    sub = {k: v[idx].contiguous() for k, v in b.items()}
    del b
    torch.cuda.empty_cache()
    out_s, grads_s = _subset_grads(model, cfg, sub, torch.arange(3, device=DEV), G)
    assert (out_b["pose_1"][idx] - out_s["pose_1"]).abs().max() <= 1e-6
    for k in grads_s:
        ref = grads_s[k]
        nrm = float((grads_b[k] - ref).norm()) / (float(ref.norm()) + 1e-20)
        assert nrm <= 2e-3, (k, nrm)
