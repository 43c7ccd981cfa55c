"""f4: Ranger.  CPU: the oracle restatement vs the reference class (golden).  GPU: the fused multi-tensor HIP step
(with nan_to_num folded in) vs the same golden after every one of 14 steps (lookahead merges at 6 and 12, the RAdam
rectification switches on at step 6), plus optimizer-state parity."""
import os

import numpy as np
import pytest
import torch

from oracle.make_golden import RANGER_SHAPES, RANGER_STEPS, ranger_problem
from oracle.ranger_oracle import clean_grad, ranger_step
from tests.util import GOLDEN_DIR

GROUPS = [dict(idx=(0, 1, 2), lr=2e-2, wd=0.0), dict(idx=(3, 4), lr=5e-3, wd=0.1)]


def _golden():
    return np.load(os.path.join(GOLDEN_DIR, "ranger_steps.npz"))


def test_ranger_oracle_matches_reference_class():
    z = _golden()
    params, grads = ranger_problem()
    states = [dict() for _ in params]
    for t in range(RANGER_STEPS):
        for grp in GROUPS:
            for i in grp["idx"]:
                params[i] = ranger_step(params[i], clean_grad(grads[t][i]), states[i], grp["lr"], weight_decay=grp["wd"])
        for i, p in enumerate(params):
            np.testing.assert_allclose(p.numpy(), z[f"p{i}_step{t + 1}"], rtol=2e-5, atol=2e-6, err_msg=f"p{i} step {t + 1}")
    for i, st in enumerate(states):
        np.testing.assert_allclose(st["exp_avg_sq"].numpy(), z[f"exp_avg_sq{i}"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(st["slow_buffer"].numpy(), z[f"slow{i}"], rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
def test_hip_ranger_matches_reference_class():
    from catre_amd.ranger import Ranger

    z = _golden()
    params, grads = ranger_problem()
    ps = [torch.nn.Parameter(p.clone().cuda()) for p in params]
    opt = Ranger([dict(params=[ps[i] for i in g["idx"]], lr=g["lr"], weight_decay=g["wd"]) for g in GROUPS], lr=1e-2,
                 clean_grads=True)
    for t in range(RANGER_STEPS):
        for p, g in zip(ps, grads[t]):
            p.grad = g.clone().cuda()   # NaN / inf left in: the fused step cleans them
        opt.step()
        for i, p in enumerate(ps):
            # measured 1.5e-7 (profiles/ranger_dev.py): the kernel rounds beta and 1 - beta separately like the reference
            np.testing.assert_allclose(p.detach().cpu().numpy(), z[f"p{i}_step{t + 1}"], rtol=2e-6, atol=2e-9,
                                       err_msg=f"p{i} step {t + 1}")
    for i, p in enumerate(ps):
        st = opt.state[p]
        assert st["step"] == RANGER_STEPS and set(st) == {"step", "exp_avg", "exp_avg_sq", "slow_buffer"}
        np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), z[f"exp_avg{i}"], rtol=3e-5, atol=1e-6)
        np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), z[f"exp_avg_sq{i}"], rtol=3e-6, atol=1e-12)
        np.testing.assert_allclose(st["slow_buffer"].cpu().numpy(), z[f"slow{i}"], rtol=2e-6, atol=2e-9)
    # state_dict round trip keeps the reference's layout
    sd = opt.state_dict()
    assert sd["param_groups"][0]["k"] == 6 and sd["param_groups"][1]["weight_decay"] == 0.1
    opt2 = Ranger([dict(params=[ps[i] for i in g["idx"]], lr=g["lr"], weight_decay=g["wd"]) for g in GROUPS], lr=1e-2)
    opt2.load_state_dict(sd)
    assert opt2.state[ps[0]]["step"] == RANGER_STEPS


@pytest.mark.gpu
def test_model_factory_builds_fused_ranger_and_it_trains():
    from catre_amd import synth
    from catre_amd.batching import batch_updater_test
    from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
    from catre_amd.config import default_cfg
    from catre_amd.ranger import Ranger

    cfg = default_cfg(num_pcl=96, num_kps=64, device="cuda:0")
    model, opt = build_model_optimizer(cfg, is_test=False)
    assert isinstance(opt, Ranger) and [len(g["params"]) for g in opt.param_groups] == [32, 28, 14]
    model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
    b = {k: v.cuda() for k, v in synth.make_inputs(4, 96, 64, seed=2).items()}
    batch_updater_test(cfg, b)
    losses = []
    for _ in range(8):
        _, ld = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                      gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
                      mean_scales=b["obj_mean_scales"], sym_info=[None] * 4, do_loss=True, cur_iter=1)
        loss = sum(ld.values())
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    # the fused step writes parameters through raw pointers (no torch version bump): the inference path must still
    # notice and re-pack its weight images
    model.eval()
    before = model.refine(b, n_iter=1)["pose_1"].clone()
    for p in model.parameters():
        p.grad = torch.ones_like(p) * 1e-3
    for _ in range(3):
        opt.step()
    after = model.refine(b, n_iter=1)["pose_1"]
    assert (after - before).abs().max() > 0, "stale packed weights after a fused Ranger step"
