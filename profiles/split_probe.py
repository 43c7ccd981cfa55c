"""'split' compute mode (catre_split.h): deviation from the fp32 reference goldens per case and per-kernel times."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import golden_names, load_golden
from tests.test_hip_parity import build_model, to_dev
worst = 0.0
for name in golden_names():
    g = load_golden(name)
    model, _ = build_model(g["cfg"], g["salt"])
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "split"
    out = model.refine(to_dev(g["batch"]), n_iter=g["K"])
    K = g["K"]
    e = max(max(np.abs(out[f"pose_{i}"].cpu().numpy() - g["ref"][f"pose_{i}"]).max(), np.abs(out[f"scale_{i}"].cpu().numpy() - g["ref"][f"scale_{i}"]).max()) for i in range(1, K + 1))
    worst = max(worst, e)
    print(json.dumps(dict(case=name, split_err=float(e))), flush=True)
print("worst", worst)
