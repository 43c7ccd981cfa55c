"""Training path: outputs and EVERY parameter gradient of one refine iteration (HIP forward + HIP backward
through torch.autograd) against torch autograd of the oracle in fp64."""
import numpy as np
import pytest
import torch

from tests.util import load_golden, recipe_sd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_grads(g, sd, Gp, Gs, dtype=torch.float64):
    from oracle import catre_oracle as O

    sdr = {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}
    b = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in g["batch"].items()}
    x, tfd = O.pose_apply(b["pcl"], b["obj_kps"], b["obj_pose_est"], b["obj_scale_est"], g["cfg"].INPUT.ZERO_CENTER_INPUT)
    pose, scale = O.model_forward(x, tfd, b["obj_pose_est"], b["obj_scale_est"], sdr, g["cfg"], K_zoom=b["K"],
                                  mean_scales=b["obj_mean_scales"])
    loss = (pose * Gp.to(dtype)).sum() + (scale * Gs.to(dtype)).sum()
    loss.backward()
    return pose.detach(), scale.detach(), {k: v.grad for k, v in sdr.items()}


@pytest.mark.parametrize("name", ["refine_b2_small", "refine_b3_ragged", "refine_b2_kpsfeat_trans", "refine_b2_noft",
                                  "refine_b2_deepim_noK"])
def test_forward_train_and_all_param_grads(name):
    from catre_amd.batching import batch_updater_test
    from catre_amd.CATRE_disR_shared import build_model_optimizer
    from catre_amd.train_forward import forward_train

    g = load_golden(name)
    cfg = g["cfg"].__deepcopy__({})
    cfg.MODEL.DEVICE = DEV
    model, _ = build_model_optimizer(cfg, is_test=False)
    sd = recipe_sd(cfg, g["salt"])
    model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=True)
    model.train()
    batch = {k: v.to(DEV) for k, v in g["batch"].items()}
    batch_updater_test(cfg, batch)
    gen = torch.Generator().manual_seed(3)
    Gp, Gs = torch.randn(g["B"], 3, 4, generator=gen), torch.randn(g["B"], 3, generator=gen)
    p = dict(model.named_parameters())
    pose, scale, aux = forward_train(p, model._opts, batch["x"], batch["tfd_kps"], batch["obj_pose_est"],
                                     batch["obj_scale_est"], batch["K"], batch["obj_mean_scales"])
    # forward agrees with the reference goldens (same bar as the fused path)
    assert np.abs(pose.detach().cpu().numpy() - g["ref"]["pose_1"]).max() <= 2e-5
    assert np.abs(scale.detach().cpu().numpy() - g["ref"]["scale_1"]).max() <= 2e-5
    ((pose * Gp.to(DEV)).sum() + (scale * Gs.to(DEV)).sum()).backward()
    rp, rs, rg = _oracle_grads(g, sd, Gp, Gs)
    unused = 0
    for k, prm in p.items():
        want = rg[k]
        if want is None:  # the never-used `norm` GroupNorms (find_unused_parameters is load-bearing)
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
            unused += 1
            continue
        assert prm.grad is not None, f"no gradient for {k}"
        got = prm.grad.cpu().double()
        scale_ref = float(want.abs().max()) + 1e-12
        err = float((got - want).abs().max()) / scale_ref
        assert err <= 2e-4, f"{name}: grad of {k}: rel-to-max error {err:.2e} (max |grad| {scale_ref:.3e})"
    assert unused == 6
