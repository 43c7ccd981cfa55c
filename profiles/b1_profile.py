"""B=1 (and B=4) K=4 refine: wall time per refine and, under rocprofv3 (profiles/prof.sh), the per-kernel breakdown of the
evaluator's operating point (one image = a handful of objects, catre_evaluator.py:292-311)."""
import json, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = default_cfg(num_pcl=1024, num_kps=1024, n_iter=4, device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
batch = {k: v.cuda() for k, v in synth.make_inputs(B, 1024, 1024, seed=3).items()}
for _ in range(5): model.refine(batch, n_iter=4)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): model.refine(batch, n_iter=4)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(json.dumps({"B": B, "K": 4, "ms_per_refine": round(dt * 1e3, 4), "us_per_iteration": round(dt / 4 * 1e6, 1)}))
