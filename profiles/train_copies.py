"""Which Python lines issue the large device-to-device copies / torch elementwise kernels of one training iteration
(torch.profiler with stacks; B=256)."""
import logging, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
from catre_amd.synth import y_axis_symmetries
from torch.profiler import profile, ProfilerActivity

B, N, M = 256, 1024, 1024
cfg = default_cfg(num_pcl=N, num_kps=M, device="cuda:0")
cfg.SOLVER.OPTIMIZER_CFG = dict(type="Ranger", lr=1e-4, weight_decay=0)
model, opt = build_model_optimizer(cfg, is_test=False)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.train()
b = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=3).items()}
sym_info = [y_axis_symmetries(314) if i % 3 == 0 else None for i in range(B)]
batch_updater_test(cfg, b)

def step():
    _, ld = model(b["x"], b["tfd_kps"], init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"], K_zoom=b["K"],
                  gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
                  mean_scales=b["obj_mean_scales"], sym_info=sym_info, do_loss=True, cur_iter=1)
    sum(ld.values()).backward(); opt.step(); opt.zero_grad(set_to_none=True)

step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dev = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    if not e.name.startswith("aten::") or dev < 40:
        continue
    if e.cpu_children and any(c.name.startswith("aten::") and (getattr(c, "device_time_total", 0) or 0) >= 40 for c in e.cpu_children):
        continue  # count leaves only
    frames = [f for f in (e.stack or []) if "catre_amd" in f or "bench" in f]
    key = (e.name, str(e.input_shapes)[:60], frames[0][-70:] if frames else "(autograd engine)")
    agg[key][0] += 1
    agg[key][1] += dev
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{v[1]/1e3:8.3f} ms  x{v[0]:<3d} {k[0]:28s} {k[1]:60s} {k[2]}")

print("--- copy-like events")
agg2 = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dev = getattr(e, "device_time_total", 0) or 0
    if dev < 40 or not any(t in e.name.lower() for t in ("copy", "memcpy", "clone", "contiguous", "fill", "zero")):
        continue
    frames = [f for f in (e.stack or []) if "catre_amd" in f]
    agg2[(e.name[:40], str(e.input_shapes)[:50], frames[0][-70:] if frames else "")][0] += 1
    agg2[(e.name[:40], str(e.input_shapes)[:50], frames[0][-70:] if frames else "")][1] += dev
for k, v in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"{v[1]/1e3:8.3f} ms  x{v[0]:<3d} {k[0]:40s} {k[1]:50s} {k[2]}")
