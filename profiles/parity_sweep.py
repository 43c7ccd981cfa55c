"""Parity statistics beyond the goldens: S seeded batches (B objects, N=M=1024, K iterations) through the HIP path in each
compute mode against the oracle (torch CPU fp32 restatement of the reference, pinned to the reference's own outputs by
tests/golden) on the same inputs and recipe weights (a fresh weight salt per seed).  Prints one JSON line per mode with
the worst and median absolute deviation of (R, t, s) after every iteration.  (A parity CHECK like the tests: the oracle
is only the checker here, nothing measured or shipped runs through it.)

usage: parity_sweep.py [S=24] [B=4] [K=4]"""
import json, logging, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
from oracle import catre_oracle as O

S, B, K = (int(v) for v in (sys.argv[1:4] + ["24", "4", "4"][len(sys.argv) - 1:]))
N = M = 1024
torch.set_num_threads(min(32, os.cpu_count() or 8))
cfg = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cuda:0")
cfg_cpu = default_cfg(num_pcl=N, num_kps=M, n_iter=K, device="cpu")
model, _ = build_model_optimizer(cfg, is_test=True)
model.eval()
dev = {m: [] for m in ("fp32", "split", "bf16")}
for seed in range(S):
    sd = synth.recipe_state_dict(expected_state_shapes(cfg), seed)  # salt = seed: different weights every time
    model.load_state_dict({k: v.cuda() for k, v in sd.items()})
    batch = synth.make_inputs(B, N, M, seed=100 + seed)
    with torch.no_grad():
        ref = O.refine_k(batch, sd, cfg_cpu, n_iter=K)
    gb = {k: v.cuda() for k, v in batch.items()}
    for mode in dev:
        model.cfg.MODEL.CATRE.COMPUTE_DTYPE = mode
        out = model.refine(gb, n_iter=K)
        d = max(float((out[f"{key}_{i}"].cpu() - ref[f"{key}_{i}"]).abs().max()) for i in range(1, K + 1) for key in ("pose", "scale"))
        dev[mode].append(d)
for mode, v in dev.items():
    print(json.dumps({"mode": mode, "batches": S, "objects_per_batch": B, "N": N, "M": M, "K": K,
                      "worst_abs_dev_vs_oracle": float(np.max(v)), "median": float(np.median(v)),
                      "p90": float(np.percentile(v, 90)), "contract": 1e-4}))
