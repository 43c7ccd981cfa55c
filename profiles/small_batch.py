"""Throughput / latency of the fused K=4 refine at the other batch sizes BASELINE.json names
(config 1: B=1, config 2: B=64 forward-only) plus the config-5 shape (N=2048, M=1024, K=8) in fp32."""
import json, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

def run(B, N, M, K, reps):
    cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
    model, _ = build_model_optimizer(cfg, is_test=True)
    sd = synth.recipe_state_dict(expected_state_shapes(cfg))
    model.load_state_dict({k: v.cuda() for k, v in sd.items()}); model.eval()
    batch = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=3).items()}
    for _ in range(5): model.refine(batch, n_iter=K)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): model.refine(batch, n_iter=K)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return {"B": B, "N": N, "M": M, "K": K, "ms_per_refine": round(dt * 1e3, 3), "object_iterations_per_s": round(B * K / dt, 1),
            "refines_per_s": round(1 / dt, 1)}

out = [run(1, 1024, 1024, 4, 200), run(16, 1024, 1024, 4, 100), run(64, 1024, 1024, 4, 50), run(256, 1024, 1024, 4, 10),
       run(256, 2048, 1024, 8, 5)]
for o in out: print(json.dumps(o))
