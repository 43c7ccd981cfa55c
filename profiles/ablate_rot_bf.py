"""Ablation of k_rot_l1_bf in the instrumented build (knob 1 bits: 4 no GELU, 8 no GN1 stats, 16 no y1 stage / stores,
32 no layer-1 sweep, 64 no layer-0 sweep): kernel time per variant - what the kernel is bound by.  Timing only.
    make -C catre_amd/csrc TRACE=1 && CATRE_HIP_LIB=$PWD/catre_amd/csrc/libcatre_hip_trace.so python profiles/ablate_rot_bf.py"""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catre_amd import hip, synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg
cfg = default_cfg(device='cuda:0')
model, _ = build_model_optimizer(cfg, is_test=True)
model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()}); model.eval()
model.cfg.MODEL.CATRE.COMPUTE_DTYPE = "bf16"
B = 256
batch = {k: v.cuda() for k, v in synth.make_inputs(B, 1024, 1024, seed=1).items()}
model.refine(batch, n_iter=2)
for knob in [int(a) for a in sys.argv[1:]] or [0, 4, 8, 16, 32, 64, 4 | 8, 4 | 8 | 16, 32 | 64, 4 | 8 | 16 | 32 | 64]:
    hip.load().catre_debug_knob(1, knob)
    model.refine(batch, n_iter=2)
    hip.profile_kernel("rot_l1", 64)
    model.refine(batch, n_iter=8)
    ms = hip.profile_collect(64)
    hip.profile_kernel(None, 0)
    ms = sorted(ms)
    print(json.dumps({"knob": knob, "k_rot_l1_bf_us_median": round(ms[len(ms) // 2] * 1e3, 1), "min": round(ms[0] * 1e3, 1), "n": len(ms)}))
hip.load().catre_debug_knob(1, 0)
