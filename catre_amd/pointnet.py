"""PointNet feature extractor of the CATRE hot path - parameter containers with HIP forwards.

Mirrors the reference's ``core/catre/models/pointnets/pointnet.py`` surface (class names,
constructor kwargs, sub-module / parameter names, default torch initialisation) so that
reference checkpoints load with ``strict=True``.  The arithmetic of ``forward`` runs in
``libcatre_hip.so``; inside ``CATRE_disR_shared.forward`` these modules are never called -
the fused refine-iteration driver reads their parameters directly.  Called on their own they take the fused
kernels when nothing needs a gradient and the layer-wise HIP training ops (``train_forward.py``) otherwise.
"""
import torch
import torch.nn as nn

from . import hip
from .runtime import HipRuntime

class STN3d(nn.Module):
    """Input transform net (reference pointnet.py:13-41)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv1d(3, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, 9)
        self.relu = nn.ReLU()
        self._rt = None

    def __getstate__(self):  # the runtime binds THIS instance's parameters: never copy / pickle it
        d = self.__dict__.copy()
        d["_rt"] = None
        return d

    def forward(self, x):
        """x [B,3,N] -> [B,3,3] (conv stack + max-pool + FC tail + identity)."""
        if self._rt is None:
            self._rt = HipRuntime(lambda: {f"pcl_net.stn.{k}": v for k, v in self.named_parameters()}, 1, 1, 1)
        if _needs_grad(self, x):
            return _stn_rows(self, x, 3)
        lib = hip.load()
        import ctypes

        B, N = x.shape[0], x.shape[2]
        pts = hip.points_desc(x, x)
        prm, packed = self._rt.params(x.device)
        ws = self._rt.workspace(B, N, 0, x.device)
        pool = torch.empty(B, 1024, dtype=torch.float32, device=x.device)
        hip.check(lib.catre_stn3d_pool(ctypes.byref(pts), prm, hip.ptr(packed), hip.ptr(pool), hip.ptr(ws), ws.numel(),
                                       B, N, 0, hip.stream_ptr(x.device)), "catre_stn3d_pool")
        named = {k: t for k, t in zip(hip.PARAM_KEYS, self._rt._param_keep)}
        return self._rt._stn_tail(pool, named, "pcl_net.stn", 3).view(B, 3, 3)


class STNkd(nn.Module):
    """Feature transform net (reference pointnet.py:44-78)."""

    def __init__(self, k=64):
        super().__init__()
        self.conv1 = nn.Conv1d(k, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k * k)
        self.relu = nn.ReLU()
        self.k = k

    def forward(self, x):
        """x [B,k,N] -> [B,k,k]: on its own the feature transform net consumes a materialised activation, so it runs
        layer by layer on the HIP training ops (inside PointNetfeat / the fused path it never sees HBM)."""
        return _stn_rows(self, x, self.k)


def _needs_grad(module, *tensors):
    return torch.is_grad_enabled() and (
        any(t.requires_grad for t in tensors if isinstance(t, torch.Tensor))
        or any(p.requires_grad for p in module.parameters())
    )


def _stn_rows(module, x, k):
    from .train_forward import _stn

    B, n = x.shape[0], x.shape[2]
    hip.require_dev_f32(x, "x", (B, k, n), contiguous=False)
    rows = x.permute(0, 2, 1).reshape(B * n, k).contiguous()
    return _stn(rows, {f"m.{name}": v for name, v in module.named_parameters()}, "m", k, B, n, 0)


class PointNetfeat(nn.Module):
    """reference pointnet.py:82-121 (BatchNorm-free variant)."""

    def __init__(self, num_points, global_feat=True, out_dim=1024, feature_transform=False, **args):
        super().__init__()
        if out_dim != 1024:
            raise NotImplementedError("the HIP trunk is built for out_dim=1024 (every shipped config)")
        self.num_points = num_points
        self.out_dim = out_dim
        self.feature_transform = feature_transform
        self.stn = STN3d()
        self.conv1 = nn.Conv1d(3, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 512, 1)
        self.conv4 = nn.Conv1d(512, out_dim, 1)
        self.global_feat = global_feat
        if self.feature_transform:
            self.fstn = STNkd(k=64)
        self._rt = None

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_rt"] = None
        return d

    def _runtime(self):
        if self._rt is None:
            self._rt = HipRuntime(lambda: {f"pcl_net.{k}": v for k, v in self.named_parameters()}, 1, 1, 1)
        return self._rt

    def forward(self, x, **args):
        """x [B,3,n] -> [B,1024] (global_feat) or [B,1088,n] = cat(global repeated, pointfeat)."""
        n_pts = x.shape[2]
        if _needs_grad(self, x):
            from .train_forward import _points_rows, pointnet_rows

            hip.require_dev_f32(x, "x", (x.shape[0], 3, n_pts), contiguous=False)
            g, pf = pointnet_rows(_points_rows(x), {f"m.{k}": v for k, v in self.named_parameters()}, x.shape[0], n_pts, 0,
                                  self.feature_transform, prefix="m")
            st = {"gfeat": g, "pointfeat": pf}
        else:
            st = self._runtime().stage_pointnet(x, None, self.feature_transform)
        g = st["gfeat"][:, : self.out_dim]
        if self.global_feat:
            return g.contiguous()
        # materialising the as-written [B,1088,n] tensor is plain data movement (pointnet.py:120-121)
        pointfeat = st["pointfeat"].view(x.shape[0], n_pts, 64).permute(0, 2, 1)
        return torch.cat([g.unsqueeze(-1).expand(-1, -1, n_pts), pointfeat], 1)
