"""Eager vs HIP-graph-replayed training iteration (catre_amd.graphed.GraphedTrainStep) at small object counts.
AMP=1 in the environment: both under torch.autocast (bf16-operand kernels, one-node rotation heads)."""
import json, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.batching import batch_updater_test
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
from catre_amd.graphed import GraphedTrainStep
from catre_amd.synth import y_axis_symmetries
AMP = os.environ.get("AMP", "0") == "1"

def run(B, N=1024, M=1024, reps=10):
    cfg = default_cfg(num_pcl=N, num_kps=M, device="cuda:0")
    sd = {k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}
    b = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=3).items()}
    batch_updater_test(cfg, b)
    sym = y_axis_symmetries(314)
    sym_info = [sym if i % 3 == 0 else None for i in range(B)]
    kw = dict(x=b["x"].contiguous(), tfd_kps=b["tfd_kps"].contiguous(), init_pose=b["obj_pose_est"], init_scale=b["obj_scale_est"],
              K_zoom=b["K"], gt_ego_rot=b["gt_rot"], gt_trans=b["gt_trans"], gt_scale=b["gt_scale"], obj_kps=b["obj_kps"],
              mean_scales=b["obj_mean_scales"])
    res = {"B": B, "autocast": AMP}
    model, opt = build_model_optimizer(cfg, is_test=False); model.load_state_dict(sd); model.train()
    def eager():
        k2 = dict(kw)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=AMP):
            out, ld = model(k2.pop("x"), k2.pop("tfd_kps"), sym_info=sym_info, do_loss=True, cur_iter=1, **k2)
        sum(ld.values()).backward(); opt.step(); opt.zero_grad(set_to_none=True)
    for _ in range(10): eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): eager()
    torch.cuda.synchronize(); res["eager_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    model2, opt2 = build_model_optimizer(cfg, is_test=False); model2.load_state_dict(sd); model2.train()
    step = GraphedTrainStep(model2, opt2, kw, sym_info, max_sym=314, amp=AMP)
    for _ in range(3): step(sym_info=sym_info, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step(sym_info=sym_info, **kw)
    torch.cuda.synchronize(); res["graphed_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    res["speedup"] = round(res["eager_ms"] / res["graphed_ms"], 2)
    return res

run(8, reps=3)  # throwaway: the first configuration of a process times its eager loop ~2x too slow
for B in ([int(v) for v in sys.argv[1:]] or (4, 8, 16, 64, 256)):
    print(json.dumps(run(B)), flush=True)
