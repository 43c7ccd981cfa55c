"""Per-phase shader-clock stamps of k_rot_l1w (instrumented build: make -C catre_amd/csrc TRACE=1;
CATRE_HIP_LIB=.../libcatre_hip_trace.so python profiles/trace_rotw.py): one wave per SIMD, a tile per wave, so a phase's
figure is that wave's own time."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catre_amd import hip, synth
from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
from catre_amd.config import default_cfg

cfg = default_cfg(device='cuda:0')
model, _ = build_model_optimizer(cfg, is_test=True)
sd = synth.recipe_state_dict(expected_state_shapes(cfg))
model.load_state_dict({k: v.cuda() for k, v in sd.items()})
model.eval()
B = 256
batch = {k: v.cuda() for k, v in synth.make_inputs(B, 1024, 1024, seed=1).items()}
model.refine(batch, n_iter=1)
tiles = B * 32
OFF = 1 << 24
buf = torch.zeros(OFF + tiles * 32, dtype=torch.int64, device='cuda')
hip.load().catre_debug_trunk_trace(ctypes.c_void_p(buf.data_ptr()))
model.refine(batch, n_iter=1)
torch.cuda.synchronize()
hip.load().catre_debug_trunk_trace(None)
t = buf[OFF:].view(tiles, 32)[:, :20].cpu().double()
t = t[2048:6144]
d = t[:, 1:] - t[:, :-1]
names = ['stage pf tile']
for h in (0, 1):
    for q in range(4):
        names += [f'h{h} q{q} layer 0 (128 asm MFMAs)', f'h{h} q{q} GELU + layer-1 slice (512 MFMAs)']
    names += [f'h{h} y1 stores + GN1 partials']
for i, nm in enumerate(names):
    print(f'  {nm:44s} {d[:, i].mean():9.0f}  (min {d[:, i].min():9.0f} max {d[:, i].max():9.0f})')
tot = (t[:, 19] - t[:, 0])
print(f'total {tot.mean():.0f} cycles per tile and wave (both heads); MFMA issue floor 2 x 2560 x 64 = {2 * 2560 * 64}')
