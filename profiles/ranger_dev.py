"""Probe: worst deviation of the fused Ranger step from the reference class's golden run (tests/golden/ranger_steps.npz).
(A parity CHECK like the tests: the oracle is only the checker here.)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.make_golden import RANGER_STEPS, ranger_problem
from catre_amd.ranger import Ranger
GROUPS = [dict(idx=(0, 1, 2), lr=2e-2, wd=0.0), dict(idx=(3, 4), lr=5e-3, wd=0.1)]
z = np.load("tests/golden/ranger_steps.npz")
params, grads = ranger_problem()
ps = [torch.nn.Parameter(p.clone().cuda()) for p in params]
opt = Ranger([dict(params=[ps[i] for i in g["idx"]], lr=g["lr"], weight_decay=g["wd"]) for g in GROUPS], lr=1e-2, clean_grads=True)
wp = 0
for t in range(RANGER_STEPS):
    for p, g in zip(ps, grads[t]):
        p.grad = g.clone().cuda()
    opt.step()
    for i, p in enumerate(ps):
        w = z[f"p{i}_step{t + 1}"]; d = np.abs(p.detach().cpu().numpy() - w) / (np.abs(w) + 1e-3)
        wp = max(wp, d.max())
def rel(a, w, fl): return float((np.abs(a - w) / (np.abs(w) + fl)).max())
print("params worst |d|/(|w|+1e-3):", wp)
for i, p in enumerate(ps):
    st = opt.state[p]
    print(i, "exp_avg", rel(st["exp_avg"].cpu().numpy(), z[f"exp_avg{i}"], 1e-6), "exp_avg_sq", rel(st["exp_avg_sq"].cpu().numpy(), z[f"exp_avg_sq{i}"], 1e-9),
          "slow", rel(st["slow_buffer"].cpu().numpy(), z[f"slow{i}"], 1e-3))
