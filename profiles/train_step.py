"""Time one training refine-iteration (forward + loss + backward + optimizer step) at a given batch size."""
import json, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
from catre_amd.synth import y_axis_symmetries

def run(B, N=1024, M=1024, reps=3):
    cfg = default_cfg(num_pcl=N, num_kps=M, device="cuda:0")
    cfg.SOLVER.OPTIMIZER_CFG = dict(type=os.environ.get("CATRE_OPT", "Ranger"), lr=1e-4, weight_decay=0)
    model, opt = build_model_optimizer(cfg, is_test=False)
    sd = synth.recipe_state_dict(expected_state_shapes(cfg))
    model.load_state_dict({k: v.cuda() for k, v in sd.items()}); model.train()
    b = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=3).items()}
    sym = y_axis_symmetries(314)
    sym_info = [sym if i % 3 == 0 else None for i in range(B)]
    batch_updater_test(cfg, b)
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(os.environ.get("CATRE_AMP"))):
            _, ld = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                          gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
                          mean_scales=b["obj_mean_scales"], sym_info=sym_info, do_loss=True, cur_iter=1)
        loss = sum(ld.values()); loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
        return loss
    step(); torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(reps): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return {"B": B, "N": N, "M": M, "ms_per_train_iteration": round(dt * 1e3, 2), "train_object_iterations_per_s": round(B / dt, 1),
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2), "loss": float(l)}

for B in ([int(v) for v in sys.argv[1:]] or (16, 64, 256)):
    print(json.dumps(run(B)))
