import sys, torch, logging
sys.path.insert(0, '/root/repo'); logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
cfg = default_cfg(device='cuda:0')
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
for B, mode in ((1500, "fp32"), (1500, "split"), (1500, "bf16"), (3000, "fp32")):
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = mode
    b = {k: v.cuda() for k, v in synth.make_inputs(B, 1024, 1024, seed=7).items()}
    out = model.refine(b, n_iter=2)
    idx = torch.tensor([0, 1, B // 2, B - 2, B - 1], device='cuda')
    sub = {k: v[idx].contiguous() for k, v in b.items()}
    o2 = model.refine(sub, n_iter=2)
    d = float((out["pose_2"][idx] - o2["pose_2"]).abs().max())
    print(B, mode, "finite", bool(torch.isfinite(out["pose_2"]).all()), "max dev vs 5-object run", d, "mem GB", round(torch.cuda.max_memory_allocated()/2**30, 1))
    del b, out
