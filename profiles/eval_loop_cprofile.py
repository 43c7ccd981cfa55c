"""cProfile of the evaluator-style loop's host side at B=1 (see eval_loop_probe.py)."""
import cProfile, pstats, logging, os, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

N = M = 1024; K = 4
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
batch = {k: v.cuda() for k, v in synth.make_inputs(1, N, M, seed=3).items()}

def loop():
    poses, scales = batch["obj_pose_est"], batch["obj_scale_est"]
    b = dict(batch)
    for it in range(1, K + 1):
        batch_updater_test(cfg, b, poses_est=poses, scales_est=scales)
        o = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                  mean_scales=b.get("obj_mean_scales"), cur_iter=it)
        poses, scales = o[f"pose_{it}"], o[f"scale_{it}"]

with torch.no_grad():
    for _ in range(20): loop()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): loop()
    pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue())
