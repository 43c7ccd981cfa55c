"""Residual heads of the CATRE hot path.  Inside ``CATRE_disR_shared.forward`` they are evaluated by the fused
driver (the concatenated feature tensors are never built); called on their own - the reference's module interface,
``heads/conv_out_per_rot_head.py:62-71,126-140`` and ``fc_trans_size_head.py:61-70`` on a materialised feature tensor -
they run layer by layer on the HIP training ops (``catre_amd/train_ops.py``), autograd included.

Mirrors ``core/catre/models/heads/conv_out_per_rot_head.py`` and ``fc_trans_size_head.py``:
same class names, constructor kwargs, ModuleList indices (``layers.0/1/3/4``, ``linears.0/1/3/4``),
never-used ``norm`` GroupNorm, and initialisation (N(0, 0.001^2) conv/linear weights, zero bias,
GN weight 1; ``fc_t``/``fc_s`` N(0, 0.01^2)), so reference checkpoints load ``strict=True``.
"""
import torch.nn as nn

def _normal_init(m, std):
    nn.init.normal_(m.weight, 0.0, std)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0.0)


def _get_norm(norm, channels, num_gn_groups):
    if norm is None or (isinstance(norm, str) and norm.lower() in ("", "none")):
        return nn.Identity()
    if norm == "GN":
        return nn.GroupNorm(num_gn_groups, channels)
    raise NotImplementedError(f"norm={norm!r}: the HIP heads implement GroupNorm ('GN'), as in every shipped config")


def _get_act(act):
    if act is not None and act.lower() == "gelu":
        return nn.GELU()
    raise NotImplementedError(f"act={act!r}: the HIP heads implement exact-erf GELU, as in every shipped config")


class RotHead(nn.Module):
    """reference conv_out_per_rot_head.py:74-140."""

    def __init__(self, in_dim=1024, feat_dim=256, num_layers=2, rot_dim=4, norm="none", num_gn_groups=32,
                 act="leaky_relu", num_classes=1, kernel_size=1, num_points=1, norm_input=False, dropout=False,
                 point_bias=True):
        super().__init__()
        if (in_dim, feat_dim, num_layers, num_classes, kernel_size, num_gn_groups) != (1088, 256, 2, 1, 1, 32) or not (
                1 <= int(rot_dim) <= 3):
            raise NotImplementedError(
                "HIP rot head is built for in_dim=1088, feat_dim=256, num_layers=2, rot_dim in {1,2,3} (3: rot6d, 2: quat), "
                "kernel_size=1, num_gn_groups=32, num_classes=1; got "
                f"{(in_dim, feat_dim, num_layers, rot_dim, num_classes, kernel_size, num_gn_groups)}"
            )
        if norm_input or dropout:
            raise NotImplementedError("norm_input / dropout are not used by the shipped configs")
        self.norm = _get_norm(norm, feat_dim, num_gn_groups)  # never used in forward (reference :92)
        self.act_func = act_func = _get_act(act)
        self.num_classes = num_classes
        self.rot_dim = rot_dim
        self.layers = nn.ModuleList()
        for i in range(num_layers):
            self.layers.append(nn.Conv1d(in_dim if i == 0 else feat_dim, feat_dim, kernel_size))
            self.layers.append(_get_norm(norm, feat_dim, num_gn_groups))
            self.layers.append(act_func)
        self.neck = nn.ModuleList([nn.Conv1d(feat_dim, rot_dim * num_classes, 1)])
        self.conv_p = nn.Conv1d(num_points, 1, 1, bias=point_bias)
        self._init_weights()

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                _normal_init(m, 0.001)
            elif isinstance(m, nn.GroupNorm):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def forward(self, x):
        """x [B,1088,P] -> (r [B,rot_dim], feat [B,rot_dim,P]) like the reference (``:126-140``): conv -> GN -> GELU ->
        conv -> GN -> GELU -> neck = feat -> conv_p over the points."""
        from . import train_ops as T

        B, C, P = x.shape
        if P != self.conv_p.in_channels:
            raise ValueError(f"RotHead was built for {self.conv_p.in_channels} points, got {P}")
        rows = x.permute(0, 2, 1).reshape(B * P, C)
        y = T.linear(rows, self.layers[0].weight, self.layers[0].bias)
        a = T.gn_points_gelu(y, self.layers[1].weight, self.layers[1].bias, B, P)
        y = T.linear(a, self.layers[3].weight, self.layers[3].bias)
        a = T.gn_points_gelu(y, self.layers[4].weight, self.layers[4].bias, B, P)
        y3 = neck_rows(a, self.neck[0].weight, self.neck[0].bias)             # [B*P,3], columns >= rot_dim are zero
        r = T.weighted_point_sum(y3, self.conv_p.weight, self.conv_p.bias, B, P)
        rd = self.rot_dim
        feat = y3.view(B, P, 3)[:, :, :rd].permute(0, 2, 1)                      # the reference's `feat = x.clone()` (:132)
        return r[:, :rd], feat  # a padded column only carries conv_p.bias: sliced away


def neck_weight3(weight, bias):
    """neck Conv1d(256 -> rot_dim) parameters as [3,256] / [3], zero-padded (differentiable) for the 3-column kernels."""
    import torch.nn.functional as F

    rd = weight.shape[0]
    w = weight.reshape(rd, -1)
    if rd < 3:
        w = F.pad(w, (0, 0, 0, 3 - rd))
        bias = F.pad(bias, (0, 3 - rd)) if bias is not None else None
    return w, bias


def neck_rows(a, weight, bias):
    """neck Conv1d(256 -> rot_dim, k=1) on point rows, zero-padded to the 3 columns the point-sum kernels are built for
    (pure data movement on [rot_dim,256] / [rot_dim]; gradients of the padding rows are dropped by autograd)."""
    import torch.nn.functional as F

    from . import train_ops as T

    rd = weight.shape[0]
    w = weight.reshape(rd, -1)
    if rd < 3:
        w = F.pad(w, (0, 0, 0, 3 - rd))
        bias = F.pad(bias, (0, 3 - rd)) if bias is not None else None
    return T.linear(a, w, bias)


class ConvOutPerRotHead(nn.Module):
    """reference conv_out_per_rot_head.py:10-71: two independent RotHeads (x axis, y axis) -> rot6d."""

    def __init__(self, in_dim=1024, feat_dim=256, num_layers=2, rot_dim=3, norm="GN", num_gn_groups=32, act="gelu",
                 num_classes=1, kernel_size=1, num_points=1, per_rot_sup=False, norm_input=False, dropout=False,
                 point_bias=True, **args):
        super().__init__()
        self.per_rot_sup = per_rot_sup
        mk = lambda: RotHead(in_dim, feat_dim, num_layers, rot_dim, norm, num_gn_groups, act, num_classes, kernel_size,
                             num_points, norm_input, dropout, point_bias)
        self.rot_head_x = mk()
        self.rot_head_y = mk()
        self.num_points = num_points
        self.rot_dim = rot_dim

    def forward(self, x):
        import torch

        rx, feat_x = self.rot_head_x(x)
        ry, feat_y = self.rot_head_y(x)
        r_pred = torch.cat((rx, ry), dim=1)            # [B, 2*rot_dim] (:63-66)
        if self.per_rot_sup:
            return r_pred, torch.cat((feat_x, feat_y), dim=1)   # (:68-69)
        return r_pred


class FC_TransSizeHead(nn.Module):
    """reference fc_trans_size_head.py:9-70."""

    def __init__(self, in_dim=1024, feat_dim=256, num_layers=2, rot_dim=4, norm="none", num_gn_groups=32,
                 act="leaky_relu", num_classes=1, norm_input=False, dropout=False):
        super().__init__()
        if (feat_dim, num_layers, num_classes, num_gn_groups) != (256, 2, 1, 32):
            raise NotImplementedError("HIP ts head is built for feat_dim=256, num_layers=2, num_gn_groups=32, num_classes=1")
        if norm_input or dropout:
            raise NotImplementedError("norm_input / dropout are not used by the shipped configs")
        self.norm = _get_norm(norm, feat_dim, num_gn_groups)  # never used in forward (reference :28)
        self.act_func = act_func = _get_act(act)
        self.num_classes = num_classes
        self.rot_dim = rot_dim
        self.in_dim = in_dim
        self.linears = nn.ModuleList()
        for i in range(num_layers):
            self.linears.append(nn.Linear(in_dim if i == 0 else feat_dim, feat_dim))
            self.linears.append(_get_norm(norm, feat_dim, num_gn_groups))
            self.linears.append(act_func)
        self.fc_t = nn.Linear(feat_dim, 3 * num_classes)
        self.fc_s = nn.Linear(feat_dim, 3 * num_classes)
        self._init_weights()

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                _normal_init(m, 0.001)
            elif isinstance(m, nn.GroupNorm):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)
        _normal_init(self.fc_t, 0.01)
        _normal_init(self.fc_s, 0.01)

    def forward(self, x):
        """x [B,in_dim] -> (trans deltas [B,3], scale deltas [B,3])."""
        from . import train_ops as T

        h = T.linear(x.flatten(1), self.linears[0].weight, self.linears[0].bias)
        h = T.gn_rows_gelu(h, self.linears[1].weight, self.linears[1].bias)
        h = T.linear(h, self.linears[3].weight, self.linears[3].bias)
        h = T.gn_rows_gelu(h, self.linears[4].weight, self.linears[4].bias)
        return T.linear(h, self.fc_t.weight, self.fc_t.bias), T.linear(h, self.fc_s.weight, self.fc_s.bias)
