"""CPU emulation of the split compute mode's arithmetic on the reference goldens - a design aid, not a test.

Before a layer is moved to the split-bf16 pipe (csrc/catre_split.h) its effect on the parity margin is predicted here:
the oracle's layers named in LAYERS are evaluated as  a_hi.w_hi + a_hi.w_lo + a_lo.w_hi  (hi = bf16(x),
lo = bf16(x - hi), products accumulated in fp32) and the worst deviation of (R, t, s) from the reference goldens over
all iterations is printed per golden.

    python tests/emulate_split.py                      # the layers the kernels split today
    python tests/emulate_split.py +trunk.conv2 +ft     # ... plus candidates
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import catre_oracle as O  # noqa: E402
from tests.util import golden_names, load_golden, recipe_sd  # noqa: E402

CURRENT = {"stn.conv2", "stn.conv3", "fstn.conv1", "fstn.conv2", "fstn.conv3", "trunk.conv3", "trunk.conv4", "rot.l0",
           "rot.l1"}
LAYERS = set(CURRENT)


def _split(t):
    hi = t.to(torch.bfloat16).to(t.dtype)
    return hi, (t - hi).to(torch.bfloat16).to(t.dtype)


def conv_x3(x, w, b=None):
    xh, xl = _split(x)
    wh, wl = _split(w)
    y = F.conv1d(xh, wh) + F.conv1d(xh, wl) + F.conv1d(xl, wh)
    return y if b is None else y + b.reshape(1, -1, 1)


def conv(name, x, w, b=None):
    return conv_x3(x, w, b) if name in LAYERS else F.conv1d(x, w, b)


def stn(x, sd, prefix, k):
    w = lambda n: sd[f"{prefix}.{n}"]
    tag = "stn" if k == 3 else "fstn"
    h = F.relu(conv(f"{tag}.conv1", x, w("conv1.weight"), w("conv1.bias")))
    h = F.relu(conv(f"{tag}.conv2", h, w("conv2.weight"), w("conv2.bias")))
    h = F.relu(conv(f"{tag}.conv3", h, w("conv3.weight"), w("conv3.bias")))
    h = torch.max(h, 2)[0]
    pooled = h
    h = F.relu(F.linear(h, w("fc1.weight"), w("fc1.bias")))
    h = F.relu(F.linear(h, w("fc2.weight"), w("fc2.bias")))
    h = F.linear(h, w("fc3.weight"), w("fc3.bias"))
    h = h + torch.eye(k, dtype=h.dtype).reshape(1, k * k)
    return h.reshape(-1, k, k), pooled


def pointnet_feat(x, sd, prefix="pcl_net", feature_transform=True, global_feat=False, detail=False):
    w = lambda n: sd[f"{prefix}.{n}"]
    n_pts = x.shape[2]
    trans, pool3 = stn(x, sd, f"{prefix}.stn", 3)
    h = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)
    h = F.relu(F.conv1d(h, w("conv1.weight"), w("conv1.bias")))
    trans_feat, pool64 = None, None
    if feature_transform:
        trans_feat, pool64 = stn(h, sd, f"{prefix}.fstn", 64)
        if "ft" in LAYERS:
            hh, hl = _split(h.transpose(2, 1))
            th, tl = _split(trans_feat)
            h = (torch.bmm(hh, th) + torch.bmm(hh, tl) + torch.bmm(hl, th)).transpose(2, 1)
        else:
            h = torch.bmm(h.transpose(2, 1), trans_feat).transpose(2, 1)
    pointfeat = h
    h = F.relu(conv("trunk.conv2", h, w("conv2.weight"), w("conv2.bias")))
    h = F.relu(conv("trunk.conv3", h, w("conv3.weight"), w("conv3.bias")))
    h = conv("trunk.conv4", h, w("conv4.weight"), w("conv4.bias"))
    g = torch.max(h, 2)[0]
    out = g if global_feat else torch.cat([g.unsqueeze(-1).repeat(1, 1, n_pts), pointfeat], 1)
    if detail:
        return out, dict(trans=trans, trans_feat=trans_feat, pointfeat=pointfeat, g=g, stn_pool=pool3, fstn_pool=pool64)
    return out


def rot_head_single(feat, sd, prefix, num_gn_groups=32):
    w = lambda n: sd[f"{prefix}.{n}"]
    w0 = w("layers.0.weight")
    h = F.conv1d(feat[:, :1024], w0[:, :1024], w("layers.0.bias")) + conv("rot.l0", feat[:, 1024:], w0[:, 1024:])
    h = F.group_norm(h, num_gn_groups, w("layers.1.weight"), w("layers.1.bias"), 1e-5)
    h = O.gelu_exact(h)
    h = conv("rot.l1", h, w("layers.3.weight"), w("layers.3.bias"))
    h = F.group_norm(h, num_gn_groups, w("layers.4.weight"), w("layers.4.bias"), 1e-5)
    h = O.gelu_exact(h)
    h = F.conv1d(h, w("neck.0.weight"), w("neck.0.bias")).permute(0, 2, 1)
    h = F.conv1d(h, w("conv_p.weight"), sd.get(f"{prefix}.conv_p.bias"))
    return h.squeeze(1).contiguous()


def main(argv):
    for a in argv:
        (LAYERS.add if a[0] == "+" else LAYERS.discard)(a[1:])
    print("split layers:", sorted(LAYERS))
    O.stn, O.pointnet_feat, O.rot_head_single = stn, pointnet_feat, rot_head_single
    worst = 0.0
    with torch.no_grad():
        for name in golden_names():
            g = load_golden(name)
            out = O.refine_k(g["batch"], recipe_sd(g["cfg"], g["salt"]), g["cfg"], n_iter=g["K"])
            err = max(np.abs(out[k].numpy() - g["ref"][k]).max() for k in g["ref"] if k.startswith(("pose_", "scale_")))
            worst = max(worst, err)
            print(f"  {name:32s} {err:.3e}")
    print(f"worst {worst:.3e}")


if __name__ == "__main__":
    main(sys.argv[1:])
