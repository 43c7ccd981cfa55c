"""ms per K=4 refine at small batch sizes (fp32): `python profiles/small_sweep.py 1 2 4 8 16 32` (CATRE_HIP_LIB selects a build)."""
import json, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

N = M = 1024
K = 4
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
sd = synth.recipe_state_dict(expected_state_shapes(cfg))
model.load_state_dict({k: v.cuda() for k, v in sd.items()}); model.eval()
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]:
    batch = {k: v.cuda() for k, v in synth.make_inputs(B, N, M, seed=3).items()}
    best = 1e9
    for rep in range(3):
        for _ in range(5): model.refine(batch, n_iter=K)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): model.refine(batch, n_iter=K)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 100)
    print(json.dumps({"B": B, "ms_per_refine": round(best * 1e3, 3), "lib": os.environ.get("CATRE_HIP_LIB", "default")}))
