"""The reference evaluator's loop as a drop-in user runs it (catre_evaluator.py:292-311): per iteration one
`batch_updater_test` + one `model(...)` call from Python - against `model.refine` (the whole K loop as one C call).
`python profiles/eval_loop_probe.py 1 2 4 6` -> wall time per K=4 refine and the host-side (launch) time of the loop."""
import json, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

N = M = 1024
K = 4
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()


def loop(batch):
    """catre_evaluator.py:292-311"""
    out = {}
    poses, scales = batch["obj_pose_est"], batch["obj_scale_est"]
    b = dict(batch)
    for it in range(1, K + 1):
        batch_updater_test(cfg, b, poses_est=poses, scales_est=scales)
        o = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                  mean_scales=b.get("obj_mean_scales"), cur_iter=it)
        poses, scales = o[f"pose_{it}"], o[f"scale_{it}"]
        out.update(o)
    return out


with torch.no_grad():
    for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 6]:
        batch = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=3).items()}
        a, r = loop(batch), model.refine(batch, n_iter=K)
        same = bool(torch.equal(a[f"pose_{K}"], r[f"pose_{K}"]) and torch.equal(a[f"scale_{K}"], r[f"scale_{K}"]))
        res = {"B": B, "K": K, "loop_equals_refine_bitwise": same}
        for name, fn in (("loop", lambda: loop(batch)), ("refine", lambda: model.refine(batch, n_iter=K))):
            for _ in range(10): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): fn()
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            res[name + "_ms"] = round((t2 - t0) / 100 * 1e3, 4)
            res[name + "_host_ms"] = round((t1 - t0) / 100 * 1e3, 4)
        print(json.dumps(res))
