"""Probe (a parity CHECK, like the tests: the oracle is only the checker here): autocast gradients of sampled objects at full
size against the rounding oracle - the development form of tests/test_hip_amp_pin.py's full-size test."""
import sys, torch
sys.path.insert(0, '/root/repo')
from catre_amd import synth
from oracle import catre_oracle as O
from tests.test_hip_fullsize import _subset_grads
from tests.test_hip_fullsize_oracle import _cfg, _sub, _train_model
DEV="cuda:0"
B, N, M = 256, 1024, 1024
model, _, cfg, sd = _train_model(N, M)
cpu = synth.make_inputs(B, N, M, seed=77)
b = {k: v.to(DEV) for k, v in cpu.items()}
idx = [0, 85, 129, 255]
gen = torch.Generator().manual_seed(2)
Gp, Gs = torch.randn(len(idx), 3, 4, generator=gen), torch.randn(len(idx), 3, generator=gen)
with torch.autocast("cuda", dtype=torch.bfloat16):
    _, grads = _subset_grads(model, cfg, b, torch.tensor(idx, device=DEV), (Gp.to(DEV), Gs.to(DEV)))
_, g32 = _subset_grads(model, cfg, b, torch.tensor(idx, device=DEV), (Gp.to(DEV), Gs.to(DEV)))
cfg_cpu = _cfg(N, M, 4)
torch.set_num_threads(16)
def emu(dtype, mode="bf16_train", sel=None):
    sdr = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
    s = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in _sub(cpu, sel or idx).items()}
    with O.operand_rounding(mode):
        x, tfd = O.pose_apply(s["pcl"], s["obj_kps"], s["obj_pose_est"], s["obj_scale_est"], True)
        rp, rs = O.model_forward(x, tfd, s["obj_pose_est"], s["obj_scale_est"], sdr, cfg_cpu, K_zoom=s["K"], mean_scales=s["obj_mean_scales"])
    ((rp * Gp.to(dtype)).sum() + (rs * Gs.to(dtype)).sum()).backward()
    return {k: v.grad for k, v in sdr.items() if v.grad is not None}
e32 = emu(torch.float32); e64 = emu(torch.float64)
print(f"{'tensor':40s} hip-e32  hip-e64  e32-e64  hip32-e32")
for k in grads:
    g = grads[k].cpu(); n = float(e32[k].norm())+1e-30
    print(f"{k:40s} {float((g-e32[k]).norm())/n:.1e} {float((g.double()-e64[k]).norm())/n:.1e} {float((e32[k].double()-e64[k]).norm())/n:.1e} {float((g32[k].cpu()-e32[k]).norm())/n:.1e}")
