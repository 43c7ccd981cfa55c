"""bf16-operand path: deviation from the fp32 reference goldens per case, and throughput at the headline shape."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import golden_names, load_golden, recipe_sd
from tests.test_hip_parity import build_model, to_dev

for name in golden_names():
    g = load_golden(name)
    cfg = g["cfg"]
    model, _ = build_model(cfg, g["salt"])
    out32 = model.refine(to_dev(g["batch"]), n_iter=g["K"])
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
    out16 = model.refine(to_dev(g["batch"]), n_iter=g["K"])
    torch.cuda.synchronize()
    K = g["K"]
    e32 = max(np.abs(out32[f"pose_{K}"].cpu().numpy() - g["ref"][f"pose_{K}"]).max(), np.abs(out32[f"scale_{K}"].cpu().numpy() - g["ref"][f"scale_{K}"]).max())
    ep1 = np.abs(out16["pose_1"].cpu().numpy() - g["ref"]["pose_1"]).max()
    epK = np.abs(out16[f"pose_{K}"].cpu().numpy() - g["ref"][f"pose_{K}"]).max()
    esK = np.abs(out16[f"scale_{K}"].cpu().numpy() - g["ref"][f"scale_{K}"]).max()
    print(json.dumps(dict(case=name, fp32_err=float(e32), bf16_pose1=float(ep1), bf16_poseK=float(epK), bf16_scaleK=float(esK))), flush=True)

from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
for (B, N, M, K) in [(256, 1024, 1024, 4), (256, 2048, 1024, 8)]:
    cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
    model, _ = build_model_optimizer(cfg, is_test=True)
    model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
    b = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=0).items()}
    for dt in ("fp32", "bf16"):
        model.cfg.MODEL.CATRE.COMPUTE_DTYPE = dt
        for _ in range(2): model.refine(b, n_iter=K)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 5
        for _ in range(reps): out = model.refine(b, n_iter=K)
        torch.cuda.synchronize(); dt_s = (time.perf_counter() - t0) / reps
        print(json.dumps(dict(B=B, N=N, M=M, K=K, dtype=dt, ms_per_refine=dt_s * 1e3, obj_it_per_s=B * K / dt_s)), flush=True)
