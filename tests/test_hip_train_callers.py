"""The caller branches of the reference's train loop that reach the module through its config surface (SURVEY.md 8b "callers
that must work unchanged"), executed on the device as the reference writes them:

* the AMP branch of ``core/catre/engine/engine.py:205-207,304,333-347`` - default-dtype ``autocast`` around forward + loss,
  ``GradScaler.scale(losses).backward()``, ``GradScaler.step(optimizer)``, ``GradScaler.update()`` - on the fused Ranger;
* ``PCLNET.FREEZE`` / ``ROT_HEAD.FREEZE`` / ``TS_HEAD.FREEZE`` (``CATRE_disR_shared.py:301-304``, ``model_utils.py:78-89,
  156-167``);
* ``SOLVER.CLIP_GRADIENTS`` (``lib/torch_utils/solver/grad_clip_d2.py:80-120``) stepping the fused Ranger on device gradients.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(B=6, N=128, M=64, seed=31, mutate=None, clean_grads=None):
    from catre_amd import synth
    from catre_amd.batching import batch_updater_test
    from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
    from catre_amd.config import default_cfg
    from oracle.catre_oracle import y_axis_symmetries

    cfg = default_cfg(num_pcl=N, num_kps=M, device=DEV)
    if clean_grads is not None:
        cfg.SOLVER.OPTIMIZER_CFG = dict(type="Ranger", lr=1e-4, weight_decay=0, clean_grads=clean_grads)
    if mutate is not None:
        mutate(cfg)
    model, opt = build_model_optimizer(cfg, is_test=False)
    model.load_state_dict({k: v.to(DEV) for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
    model.train()
    b = {k: v.to(DEV) for k, v in synth.make_inputs(B, N, M, seed=seed).items()}
    batch_updater_test(cfg, b)
    sym = [y_axis_symmetries(12) if i % 3 == 0 else None for i in range(B)]
    return cfg, model, opt, b, sym


def _forward(model, b, sym, cur_iter=1):
    return model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                 obj_class=b.get("obj_cls"), gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"],
                 obj_kps=b["obj_kps"], mean_scales=b["obj_mean_scales"], sym_info=sym, do_loss=True, cur_iter=cur_iter)


# ------------------------------------------------------------------------------------------------ AMP branch
def test_reference_amp_branch_runs_verbatim_on_the_fused_ranger():
    """engine.py:304,333-347 as written: ``with autocast(enabled=True)`` (default dtype: fp16 on "cuda"), ``GradScaler()``
    (initial scale 65536), ``scale(losses).backward(); step(optimizer); update()``.  Any autocast request selects the
    bf16-operand kernels with fp32 accumulation and fp32 outputs, and every backward kernel is linear in the incoming
    gradient, so the power-of-two loss scale is exact: the losses equal the bf16-autocast run's bit for bit and the
    parameters after 4 scaled steps equal 4 plain ``backward(); step()`` iterations under bf16 autocast."""
    from torch.cuda.amp import GradScaler, autocast

    cfg, model, opt, b, sym = _setup()
    cfg2, model2, opt2, _, _ = _setup()
    grad_scaler = GradScaler()
    assert grad_scaler.get_scale() == 65536.0
    losses_a, losses_b = [], []
    for refine_i in range(1, 5):
        # -- the reference branch, verbatim (AMP_ON = True)
        with autocast(enabled=True):
            out_dict, loss_dict = _forward(model, b, sym, refine_i)
            losses = sum(loss_dict.values())
            assert torch.isfinite(losses).all(), loss_dict
        assert out_dict[f"pose_{refine_i}"].dtype == torch.float32 and losses.dtype == torch.float32
        grad_scaler.scale(losses).backward()
        grad_scaler.step(opt)
        grad_scaler.update()
        opt.zero_grad(set_to_none=True)
        losses_a.append({k: v.detach().clone() for k, v in loss_dict.items()})
        # -- the unscaled run: bf16 autocast, plain backward / step
        with torch.autocast("cuda", dtype=torch.bfloat16):
            _, ld2 = _forward(model2, b, sym, refine_i)
            l2 = sum(ld2.values())
        l2.backward()
        opt2.step()
        opt2.zero_grad(set_to_none=True)
        losses_b.append({k: v.detach().clone() for k, v in ld2.items()})
    assert grad_scaler.get_scale() == 65536.0, "no step may have been skipped"
    for la, lb in zip(losses_a, losses_b):
        assert set(la) == set(lb)
        for k in la:
            assert torch.equal(la[k], lb[k]), (k, float(la[k]), float(lb[k]))
    moved = 0
    for (k, p), (_, q) in zip(model.named_parameters(), model2.named_parameters()):
        den = float(q.detach().abs().max()) + 1e-30
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-6 * den, k
        moved += int(opt.state[p]["step"] == 4) if p in opt.state else 0
    assert moved == 68


@pytest.mark.parametrize("clean_grads", [False, True])
def test_grad_scaler_skips_the_step_on_an_injected_inf(clean_grads):
    """An ``inf`` gradient must make ``GradScaler`` skip the optimizer step and halve the scale - also when the fused Ranger
    was built with ``clean_grads=True`` (the non-AMP branch's ``nan_to_num(..., posinf=1e5)``, engine.py:351-353, folded into
    the step kernel).  The scaler wins: ``GradScaler.step`` unscales and checks the ``.grad`` tensors BEFORE it calls
    ``optimizer.step()``, so the clamp inside the step kernel never sees - and cannot hide - the inf.  (The reference's AMP
    branch does not run ``nan_to_num`` at all.)"""
    from torch.cuda.amp import GradScaler, autocast

    cfg, model, opt, b, sym = _setup(clean_grads=clean_grads)
    assert opt.clean_grads is clean_grads
    grad_scaler = GradScaler()
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    with autocast(enabled=True):
        _, loss_dict = _forward(model, b, sym)
        losses = sum(loss_dict.values())
    grad_scaler.scale(losses).backward()
    victim = model.rot_head.rot_head_x.layers[3].weight
    victim.grad[7, 11] = float("inf")
    grad_scaler.step(opt)
    grad_scaler.update()
    assert grad_scaler.get_scale() == 32768.0, "the scaler must have seen the inf"
    for k, p in model.named_parameters():
        assert torch.equal(p.detach(), before[k]), f"{k} moved although the step had to be skipped"
    assert all(len(opt.state[p]) == 0 for g in opt.param_groups for p in g["params"]), "optimizer state advanced"
    opt.zero_grad(set_to_none=True)
    # the next iteration (finite gradients, scale 32768) steps normally
    with autocast(enabled=True):
        _, loss_dict = _forward(model, b, sym)
        losses = sum(loss_dict.values())
    grad_scaler.scale(losses).backward()
    grad_scaler.step(opt)
    grad_scaler.update()
    assert grad_scaler.get_scale() == 32768.0
    assert sum(int(not torch.equal(p.detach(), before[k])) for k, p in model.named_parameters()) == 68


# ------------------------------------------------------------------------------------------------ FREEZE
_FREEZE = {"PCLNET": "pcl_net.", "ROT_HEAD": "rot_head.", "TS_HEAD": "ts_head."}


@pytest.mark.parametrize("shape", [(6, 128, 64), (5, 96, 40)])   # fused-kernel shapes / ragged (layer-wise encoder, per-head rot)
@pytest.mark.parametrize("which", ["PCLNET", "ROT_HEAD", "TS_HEAD"])
def test_freeze_flags(which, shape):
    """``<PART>.FREEZE`` (CATRE_disR_shared.py:301-304, model_utils.py:78-82,156-160): the part's parameters have
    ``requires_grad=False`` and no param group; after a training iteration they hold no gradient, every other gradient is
    bit-equal to the unfrozen run's, and the optimizer steps the trainable tensors only."""
    B, N, M = shape
    cfg0, model0, opt0, b, sym = _setup(B, N, M)

    def mutate(cfg):
        cfg.MODEL.CATRE[which].FREEZE = True

    cfg, model, opt, _, _ = _setup(B, N, M, mutate=mutate)
    prefix = _FREEZE[which]
    frozen = [k for k, p in model.named_parameters() if not p.requires_grad]
    assert frozen and all(k.startswith(prefix) for k in frozen)
    assert all(not p.requires_grad for k, p in model.named_parameters() if k.startswith(prefix))
    # param groups: pcl_net @ BASE_LR, rot_head and ts_head @ BASE_LR x LR_MULT, the frozen part's group missing
    want = {"PCLNET": [28, 14], "ROT_HEAD": [32, 14], "TS_HEAD": [32, 28]}[which]
    assert [len(g["params"]) for g in opt.param_groups] == want
    assert [len(g["params"]) for g in opt0.param_groups] == [32, 28, 14]
    lr0 = float(cfg.SOLVER.BASE_LR)
    lrs = {"pcl_net.": lr0, "rot_head.": lr0 * cfg.MODEL.CATRE.ROT_HEAD.get("LR_MULT", 1.0),
           "ts_head.": lr0 * cfg.MODEL.CATRE.TS_HEAD.get("LR_MULT", 1.0)}
    names = {id(p): k for k, p in model.named_parameters()}
    for g in opt.param_groups:
        pre = {names[id(p)].split(".")[0] + "." for p in g["params"]}
        assert len(pre) == 1 and abs(g["lr"] - lrs[pre.pop()]) < 1e-12

    for m_ in (model0, model):
        _, ld = _forward(m_, b, sym)
        sum(ld.values()).backward()
    g0 = {k: p.grad for k, p in model0.named_parameters()}
    n_live = 0
    for k, p in model.named_parameters():
        if k.startswith(prefix) or g0[k] is None:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        if which == "PCLNET":
            # the frozen encoder runs on the fused inference kernels (no saves) whatever the shape; the unfrozen run's
            # encoder is the SAVE instances (64-grid shapes) or one row GEMM per layer (ragged): fp32 re-association only
            err = float((p.grad - g0[k]).abs().max()) / (float(g0[k].abs().max()) + 1e-30)
            assert err <= 2e-4, (k, err)
        else:
            assert torch.equal(p.grad, g0[k]), (k, float((p.grad - g0[k]).abs().max()))
        n_live += 1
    assert n_live == {"PCLNET": 36, "ROT_HEAD": 44, "TS_HEAD": 56}[which]
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    for k, p in model.named_parameters():
        changed = not torch.equal(p.detach(), before[k])
        assert changed == (p.grad is not None), k


def test_frozen_pcl_net_takes_the_inference_encoder_kernels(monkeypatch):
    """With ``PCLNET.FREEZE`` the training forward must not run the SAVE instances of the encoder kernels (+0.5 ms and
    +7.8 GB of activation stores per iteration at B = 256) and must not build encoder graph nodes: the encoder entry points of
    the training path are never called, the pooled feature carries no ``grad_fn``."""
    from catre_amd import runtime, train_ops

    def mutate(cfg):
        cfg.MODEL.CATRE.PCLNET.FREEZE = True

    cfg, model, opt, b, sym = _setup(4, 128, 64, mutate=mutate)
    calls = []
    for name in ("train_stn3d", "train_stnkd", "train_trunk"):
        monkeypatch.setattr(runtime.HipRuntime, name, lambda *a, _n=name, **k: calls.append(_n))
    for name in ("pooled_chain", "pointfeat_hub", "linear_maxpool"):
        monkeypatch.setattr(train_ops, name, lambda *a, _n=name, **k: calls.append(_n))
    staged = []
    orig = runtime.HipRuntime.stage_pointnet
    monkeypatch.setattr(runtime.HipRuntime, "stage_pointnet", lambda self, *a, **k: staged.append(1) or orig(self, *a, **k))
    out, ld = _forward(model, b, sym)
    sum(ld.values()).backward()
    assert calls == [] and staged == [1]
    assert all(p.grad is None for k, p in model.named_parameters() if k.startswith("pcl_net."))
    # forward parity of this path: the fused inference kernels on the same weights
    model.eval()
    with torch.no_grad():
        want = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                     mean_scales=b["obj_mean_scales"], cur_iter=1)
    assert (out["pose_1"].detach() - want["pose_1"]).abs().max() <= 2e-5


def test_frozen_pcl_net_forward_sees_out_of_band_head_writes():
    """with ``PCLNET.FREEZE`` the training forward starts on the inference encoder kernels, not on
    ``train_stn3d`` - it must still re-pack the images it reads: head weights rewritten through ``p.data`` (the reference's
    own ``lib/torch_utils/solver/ranger.py`` does ``p.data.copy_``; EMA too) bump neither ``_version`` nor the parameter
    epoch, and the heads' backward reads the live w0 / w1.  Losses and gradients must equal, bit for bit, a second model that
    got the same weights through ``load_state_dict``."""
    def mutate(cfg):
        cfg.MODEL.CATRE.PCLNET.FREEZE = True

    cfg, m_a, o_a, b, sym = _setup(4, 128, 64, mutate=mutate)
    _, m_b, o_b, _, _ = _setup(4, 128, 64, mutate=mutate)
    _, ld = _forward(m_a, b, sym)
    sum(ld.values()).backward()
    o_a.zero_grad(set_to_none=True)
    new = {k: v.detach() * 1.01 for k, v in m_a.state_dict().items()}
    for k, p in m_a.named_parameters():
        p.data.copy_(new[k])
    m_b.load_state_dict(new)
    res = []
    for m_ in (m_a, m_b):
        _, ld = _forward(m_, b, sym)
        sum(ld.values()).backward()
        res.append({k: v.detach().clone() for k, v in ld.items()})
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
    n = 0
    for (k, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert (pa.grad is None) == (pb.grad is None) and (pa.grad is None or torch.equal(pa.grad, pb.grad)), k
        n += pa.grad is not None
    assert n > 0


# ------------------------------------------------------------------------------------------------ CLIP_GRADIENTS
@pytest.mark.parametrize("clip_type", ["full_model", "norm", "value"])
def test_clip_gradients_steps_the_fused_ranger_on_device_gradients(clip_type):
    """``SOLVER.CLIP_GRADIENTS.ENABLED`` (grad_clip_d2.py:80-120) wraps ``step``: clip on the device gradients the HIP backward
    produced, then the fused step.  Checked against (i) torch's own clip functions applied to copies of those gradients
    followed by an UNCLIPPED fused Ranger on a twin model - bit-equal parameters - and (ii) the reference Ranger's arithmetic
    (``oracle.ranger_oracle``, pinned to the reference class by ``ranger_steps.npz``) on the clipped gradients."""
    from oracle.ranger_oracle import ranger_step

    clip_value = {"full_model": 0.05, "norm": 0.01, "value": 1e-3}[clip_type]

    def mutate(cfg):
        cfg.SOLVER.CLIP_GRADIENTS = dict(ENABLED=True, CLIP_TYPE=clip_type, CLIP_VALUE=clip_value, NORM_TYPE=2.0)

    cfg, model, opt, b, sym = _setup(mutate=mutate)
    cfg2, twin, opt2, _, _ = _setup()
    assert type(opt).__name__ == "RangerWithGradientClip" and type(opt2).__name__ == "Ranger"
    p0 = {k: p.detach().cpu().double() for k, p in model.named_parameters()}
    for m_ in (model, twin):
        _, ld = _forward(m_, b, sym)
        sum(ld.values()).backward()
    raw = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    # (i) torch's clip on the twin's (identical) gradients, then the plain fused step
    tp = [p for p in twin.parameters() if p.grad is not None]
    if clip_type == "full_model":
        total = torch.nn.utils.clip_grad_norm_(tp, clip_value, 2.0)
        assert float(total) > clip_value, "the clip must be active for this test to mean anything"
    elif clip_type == "norm":
        assert any(float(p.grad.norm()) > clip_value for p in tp)
        for p in tp:
            torch.nn.utils.clip_grad_norm_(p, clip_value, 2.0)
    else:
        assert any(float(p.grad.abs().max()) > clip_value for p in tp)
        for p in tp:
            torch.nn.utils.clip_grad_value_(p, clip_value)
    clipped = {k: p.grad.clone() for k, p in twin.named_parameters() if p.grad is not None}
    assert any(not torch.equal(clipped[k], raw[k]) for k in raw)
    opt2.step()
    opt.step()
    for (k, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), k
        if k in raw:   # the wrapped step clipped the model's own .grad in place, like the reference's
            assert torch.equal(p.grad, clipped[k]), k
    # (ii) the reference Ranger's update on the clipped gradients (fp64 arithmetic, fp32 storage)
    lr0 = float(cfg.SOLVER.BASE_LR)
    for k, p in model.named_parameters():
        if k not in raw:
            continue
        lr = lr0 if k.startswith("pcl_net.") else lr0 * cfg.MODEL.CATRE[("ROT_HEAD" if k.startswith("rot_head.") else "TS_HEAD")].get("LR_MULT", 1.0)
        want = ranger_step(p0[k], clipped[k].cpu().double(), {}, lr, storage=torch.float32)
        got = p.detach().cpu().double()
        step = float((want - p0[k]).abs().max())
        # (one fp32 ulp of the stored parameter on top: tiny clipped steps are of that order)
        assert float((got - want).abs().max()) <= 2e-3 * step + 1.2e-7 * float(p0[k].abs().max()) + 1e-12, (k, step)
