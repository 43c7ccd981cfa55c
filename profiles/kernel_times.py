"""Per-kernel average launch time of one refine iteration, measured with the library's in-stream HIP events
(catre_profile_enable) - a quick alternative to a rocprofv3 run.  usage: kernel_times.py [fp32|bf16] [B N M K]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catre_amd import hip, synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B, N, M, K = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (256, 1024, 1024, 4)
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
model.cfg.MODEL.CATRE.COMPUTE_DTYPE = dt
b = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=0).items()}
for _ in range(2): model.refine(b, n_iter=K)
res = {}
for name in ("stn3d", "stnkd", "trunk", "ts_head", "rot_l0_stats", "rot_l1", "rot_out"):
    hip.profile_kernel(name, 3 * K)
    for _ in range(3): model.refine(b, n_iter=K)
    ms = hip.profile_collect(3 * K); hip.profile_kernel(None, 0)
    res[name] = round(sum(ms) / len(ms), 4) if ms else None
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5): model.refine(b, n_iter=K)
torch.cuda.synchronize()
res["iteration_total"] = round((time.perf_counter() - t0) / 5 / K * 1e3, 4)
res["dtype"] = dt; res["shape"] = [B, N, M, K]
print(json.dumps(res))
