"""Which torch (aten) kernels still run inside a training iteration, and from where: torch.profiler with stacks."""
import logging, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
from catre_amd.synth import y_axis_symmetries
from torch.profiler import profile, ProfilerActivity

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = default_cfg(device="cuda:0")
cfg.SOLVER.OPTIMIZER_CFG = dict(type="Ranger", lr=1e-5, weight_decay=0, clean_grads=True)
model, opt = build_model_optimizer(cfg, is_test=False)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.train()
b = {k: v.cuda() for k, v in synth.make_inputs(B, 1024, 1024, seed=3).items()}
sym = y_axis_symmetries(314)
sym_info = [sym if i % 3 == 0 else None for i in range(B)]
batch_updater_test(cfg, b)
def step():
    _, ld = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                  gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
                  mean_scales=b["obj_mean_scales"], sym_info=sym_info, do_loss=True, cur_iter=1)
    sum(ld.values()).backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    stack = [s for s in e.stack if "catre_amd" in s or "bench" in s or "profiles" in s][:3]
    print(f"{e.key:28s} n={e.count:3d} dev_us={e.device_time_total:9.1f} shapes={str(e.input_shapes)[:70]:70s} | {' <- '.join(s.split('/')[-1][:60] for s in stack)}")
