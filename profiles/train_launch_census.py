"""One training iteration at B = 256: (1) every C-ABI call the Python side makes, by entry point and calling line;
(2) every aten op that launched a device kernel, with the calling line.  AMP=1: under torch.autocast(bf16).
    python profiles/train_launch_census.py [B]"""
import collections, logging, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import hip, synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
from catre_amd.synth import y_axis_symmetries
from torch.profiler import profile, ProfilerActivity

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
AMP = os.environ.get("AMP") == "1"
cfg = default_cfg(device="cuda:0")
cfg.SOLVER.OPTIMIZER_CFG = dict(type="Ranger", lr=1e-5, weight_decay=0, clean_grads=True)
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


for _ in range(2):
    step()
torch.cuda.synchronize()

# (1) C-ABI calls
lib = hip.load()
calls = collections.Counter()
orig = {}
for name in hip.EXPORTED_SYMBOLS:
    f = getattr(lib, name)
    orig[name] = f

    def wrap(*a, _f=f, _n=name):
        fr = [x for x in traceback.extract_stack(limit=8)[:-1] if "catre_amd" in x.filename]
        site = " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in reversed(fr[-3:]))
        calls[(_n, site)] += 1
        return _f(*a)
    setattr(lib, name, wrap)
step(); torch.cuda.synchronize()
for name, f in orig.items():
    setattr(lib, name, f)
skip = ("_ws_bytes", "catre_form", "catre_profile", "catre_param_epoch", "catre_last_error")
print(f"--- C-ABI calls of one iteration (AMP={int(AMP)}, B={B}): {sum(v for (n, _), v in calls.items() if not any(s in n for s in skip))}")
for (n, site), v in sorted(calls.items(), key=lambda kv: (-kv[1], kv[0])):
    if not any(s in n for s in skip):
        print(f"x{v:<3d} {n:40s} {site}")

# (2) aten ops with device time
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dev = getattr(e, "self_device_time_total", 0) or 0
    if not e.name.startswith("aten::") or dev <= 0:
        continue
    frames = [f for f in (e.stack or []) if "catre_amd" in f or "bench" in f or "profiles" in f]
    key = (e.name, str(e.input_shapes)[:56], " <- ".join(f.split("/")[-1][:48] for f in frames[:2]) if frames else "(autograd engine)")
    agg[key][0] += 1
    agg[key][1] += dev
print(f"--- aten ops with device kernels: {sum(v[0] for v in agg.values())} launches, {sum(v[1] for v in agg.values()):.0f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:7.1f} us x{v[0]:<3d} {k[0]:24s} {k[1]:56s} {k[2]}")
