"""CPU emulation of cheaper erf-GELU forms on the reference goldens - a design aid, not a test (like emulate_split.py).

The rotation heads evaluate 2 x 2 x 256 GELUs per point; `k_rot_out` / `k_rot_l1` pay ~18 VALU issue slots for each
(clamp, 7-coefficient numerator, 5-coefficient denominator, v_rcp: `erf_rational` in csrc/catre_device.h, 4.5e-7).  Before
a cheaper form goes into a kernel its effect on the parity margin is predicted here: the oracle's GELU is replaced by
x * P_a(x^2) / Q_b(x^2) forms fitted below (evaluated in fp32 like the kernel would) and the worst deviation of
(R, t, s) from the reference goldens over all iterations is printed.

    python tests/emulate_gelu.py            # fits (a, b) in {(6,4) shipped, (5,4), (4,4), (4,3), (3,3)} and evaluates them
"""
import os
import sys

import numpy as np
import torch
from scipy.optimize import least_squares
from scipy.special import erf

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import catre_oracle as O  # noqa: E402
from tests.util import golden_names, load_golden, recipe_sd  # noqa: E402


def fit(a, b, clamp):
    """erf(z) ~ z P_a(t) / Q_b(t), t = z^2, |z| <= clamp; weighted towards the GELU error 0.5 x erf_err, minimax by
    iterative re-weighting."""
    z = np.linspace(1e-4, clamp, 6001)
    t = z * z

    def model(p, zz, tt):
        P = sum(p[k] * tt ** k for k in range(a + 1))
        Q = 1 + sum(p[a + 1 + k] * tt ** (k + 1) for k in range(b))
        return zz * P / Q

    w = np.maximum(z, 0.02)
    wt = np.ones_like(z)
    p = np.zeros(a + 1 + b)
    p[0] = 2 / np.sqrt(np.pi)
    for _ in range(25):
        r = least_squares(lambda q: (model(q, z, t) - erf(z)) * w * wt, p, method="lm", max_nfev=4000)
        p = r.x
        e = np.abs((model(p, z, t) - erf(z)) * w)
        wt = wt * (1 + 1.5 * e / e.max())
        wt /= wt.mean()
    return p


def make_gelu(p, a, b, clamp):
    pc = [np.float32(v) for v in p]

    def gelu(v):
        v32 = v.float()
        z = torch.clamp(v32 * np.float32(0.70710678118654752440), -clamp, clamp)
        t = z * z
        P = torch.full_like(t, float(pc[a]))
        for k in range(a - 1, -1, -1):
            P = torch.addcmul(torch.full_like(t, float(pc[k])), t, P)
        Q = torch.full_like(t, float(pc[a + b]))
        for k in range(b - 2, -1, -1):
            Q = torch.addcmul(torch.full_like(t, float(pc[a + 1 + k])), t, Q)
        Q = torch.addcmul(torch.ones_like(t), t, Q)
        er = z * P / Q
        hv = 0.5 * v32
        return (hv * er + hv).to(v.dtype)
    return gelu


def gelu_err(g):
    x = torch.linspace(-8, 8, 400001, dtype=torch.float64)
    ref = 0.5 * x * (1 + torch.erf(x / np.sqrt(2)))
    return float((g(x.float()).double() - ref).abs().max())


def main():
    cands = []
    for (a, b) in ((5, 4), (4, 4), (4, 3), (3, 3), (3, 2)):
        best = None
        for clamp in (3.0, 3.2, 3.4, 3.6, 3.8, 4.0):
            p = fit(a, b, clamp)
            g = make_gelu(p, a, b, np.float32(clamp))
            e = gelu_err(g)
            if best is None or e < best[0]:
                best = (e, clamp, p, g)
        cands.append(((a, b), best))
        print(f"P{a}/Q{b}: GELU max abs error {best[0]:.2e} (clamp {best[1]}), {a + b} FMAs + rcp; coefficients {[float(np.float32(v)) for v in best[2]]}")
    exact = O.gelu_exact
    for (a, b), (e, clamp, p, g) in cands:
        worst = 0.0
        for name in golden_names():
            gd = load_golden(name)
            sd = recipe_sd(gd["cfg"], gd["salt"])
            O.gelu_exact = g
            try:
                with torch.no_grad():
                    out = O.refine_k(gd["batch"], sd, gd["cfg"], n_iter=gd["K"])
            finally:
                O.gelu_exact = exact
            for i in range(1, gd["K"] + 1):
                for key in (f"pose_{i}", f"scale_{i}"):
                    worst = max(worst, float(np.abs(out[key].numpy() - gd["ref"][key]).max()))
        print(f"P{a}/Q{b}: worst deviation of (R, t, s) from the reference goldens, all iterations: {worst:.2e} (bar 2e-5, contract 1e-4)")


if __name__ == "__main__":
    main()
