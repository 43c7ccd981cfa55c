"""Host time to ENQUEUE one training iteration against the GPU time to run it (B = 256 unless given): the loop is timed
without a device sync inside, then to the sync.  AMP=1: under torch.autocast(bf16).
    python profiles/train_host_time.py [B] [iterations]"""
import json, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
from catre_amd.synth import y_axis_symmetries

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
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


for _ in range(3):
    step()
torch.cuda.synchronize()
# host alone: a tiny batch keeps the GPU out of the way?  No - same B: the queue is deep enough not to block the host
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"B": B, "amp": int(AMP), "iterations": n, "host_enqueue_ms_per_iteration": round((t1 - t0) / n * 1e3, 3),
                  "wall_ms_per_iteration": round((t2 - t0) / n * 1e3, 3)}))
