"""Where the HOST spends a training iteration (cProfile of 20 iterations at a small batch, where the GPU is not the bound).
    [AMP=1] python profiles/train_cprofile.py [B]"""
import cProfile, io, logging, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
from catre_amd.synth import y_axis_symmetries

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
AMP = os.environ.get("AMP") == "1"
cfg = default_cfg(device="cuda:0")
cfg.SOLVER.OPTIMIZER_CFG = dict(type="Ranger", lr=1e-5, weight_decay=0)
model, opt = build_model_optimizer(cfg, is_test=False)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.train()
b = {k: v.cuda() for k, v in synth.make_inputs(B, 1024, 1024, seed=3).items()}
sym = y_axis_symmetries(314)
sym_info = [sym if i % 3 == 0 else None for i in range(B)]
batch_updater_test(cfg, b)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=AMP):
        _, ld = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                      gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
                      mean_scales=b["obj_mean_scales"], sym_info=sym_info, do_loss=True, cur_iter=1)
    sum(ld.values()).backward(); opt.step(); opt.zero_grad(set_to_none=True)


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:45]))

# the autograd worker thread (where the custom Functions' backward() bodies run): a profiler enabled from inside it
import catre_amd.losses as _L
bpr = cProfile.Profile()
_orig = _L._FusedLoss.backward
_on = [False]


def _hooked(ctx, *a):
    if not _on[0]:
        _on[0] = True
        bpr.enable()          # (this thread's profile hook)
    return _orig(ctx, *a)


_L._FusedLoss.backward = staticmethod(_hooked)
for _ in range(21):
    step()
torch.cuda.synchronize()
bpr.create_stats()
s = io.StringIO()
pstats.Stats(bpr, stream=s).sort_stats("cumulative").print_stats(45)
print("--- autograd worker thread, 20+ iterations")
print("\n".join(l[:160] for l in s.getvalue().splitlines()[:70]))
