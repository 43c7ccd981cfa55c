"""Experiment (instrumented library, `make -C catre_amd/csrc TRACE=1`, CATRE_HIP_LIB=.../libcatre_hip_trace.so): do the two
co-resident workgroups of the STN kernels lose time because they run their prologues in lock step?  Starts the odd wave
slot of the first dispatch round `d` cycles late and times the kernels."""
import json, logging, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.disable(logging.CRITICAL)
from catre_amd import hip, synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

cfg = default_cfg(device="cuda:0")
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
b = {k: v.cuda() for k, v in synth.make_inputs(256, 1024, 1024, seed=0).items()}
lib = hip.load()
for d in (0, 10000, 20000, 40000, 60000):
    hip.check(lib.catre_debug_knob(0, d), "knob")
    for _ in range(2): model.refine(b, n_iter=4)
    res = {"dephase_cycles": d}
    for name in ("stn3d", "stnkd"):
        hip.profile_kernel(name, 12)
        for _ in range(3): model.refine(b, n_iter=4)
        ms = hip.profile_collect(12); hip.profile_kernel(None, 0)
        res[name + "_ms"] = round(sum(ms) / len(ms), 4)
    print(json.dumps(res))
