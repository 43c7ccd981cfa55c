import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import load_golden
from tests.test_hip_parity import build_model, to_dev
from oracle import catre_oracle as O
for name in ["refine_b2_small", "refine_b2_n1024", "refine_b2_noft"]:
    g = load_golden(name)
    model, sd = build_model(g["cfg"], g["salt"])
    model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
    out = model.refine(to_dev(g["batch"]), n_iter=g["K"])
    with O.operand_rounding("bf16"):
        emu = O.refine_k(g["batch"], sd, g["cfg"], n_iter=g["K"])
    for i in range(1, g["K"] + 1):
        R = np.abs(out[f"pose_{i}"].cpu().numpy()[:, :, :3] - emu[f"pose_{i}"].numpy()[:, :, :3]).max()
        t = np.abs(out[f"pose_{i}"].cpu().numpy()[:, :, 3] - emu[f"pose_{i}"].numpy()[:, :, 3]).max()
        s = np.abs(out[f"scale_{i}"].cpu().numpy() - emu[f"scale_{i}"].numpy()).max()
        print(name, i, "R %.2e t %.2e s %.2e" % (R, t, s), flush=True)
