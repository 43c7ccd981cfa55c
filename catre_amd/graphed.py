"""One training refine-iteration - forward, loss, backward, optimizer step - captured in a HIP graph.

The training path launches ~450 kernels per iteration, and everything in the iteration is shape-static and free of
host synchronisation (device-side loss, fused optimizer), so it can be captured once and replayed:

    step = GraphedTrainStep(model, optimizer, example_batch, sym_info)       # warm-up + capture
    out_dict, loss_dict = step(x=..., tfd_kps=..., init_pose=..., ..., sym_info=[...])

Per call the host only copies the batch into the static input buffers, advances the optimizer's scalar state
(``Ranger.prepare_step`` -> a pinned table that is uploaded right before the replay) and replays the graph.  The
outputs are static tensors that the next call overwrites (``clone()`` what must survive).  Shapes (B, N, M), the
loss configuration and the maximum number of symmetry candidates are fixed at capture time; the mix of symmetric /
non-symmetric objects may change freely (it lives in device tensors).

This is an opt-in wrapper around the same module, kernels and optimizer - the reference's eager train loop
(``core/catre/engine/engine.py:293-355``) keeps working unchanged.  Replays are bit-identical to the eager loop
(``tests/test_hip_train.py::test_graphed_train_step_replays_the_eager_iteration``).

Measured (``profiles/train_step_graphed.py``, one MI355X, N=M=1024): the eager loop has a host-side floor of
~4.85 ms per iteration (B <= 16: ~450 launches); replaying takes 3.5 / 3.9 / 4.7 ms at B = 4 / 8 / 16 (1.4x / 1.26x /
1.04x) and is on par with the eager loop from B = 32 up, where the GPU work itself (a serial chain of short kernels)
is the bound.  Besides the small-batch gain the graph removes the per-iteration host CPU load.
"""
import torch

from . import hip
from .losses import SymTensors
from .ranger import Ranger

_INPUTS = ("x", "tfd_kps", "init_pose", "init_scale", "K_zoom", "gt_ego_rot", "gt_trans", "gt_scale", "obj_kps",
           "mean_scales")


class GraphedTrainStep:
    def __init__(self, model, optimizer, example, sym_info, max_sym=None, warmup=3, amp=False):
        if not isinstance(optimizer, Ranger):
            raise TypeError("GraphedTrainStep needs the fused catre_amd.ranger.Ranger (its step is capturable)")
        self.model, self.opt, self.amp = model, optimizer, bool(amp)
        dev = example["x"].device
        self.static = {k: example[k].detach().clone().contiguous() for k in _INPUTS if example.get(k) is not None}
        B = self.static["x"].shape[0]
        if max_sym is None:
            max_sym = max([1] + [len(s) for s in sym_info if s is not None])
        self.s1 = int(max_sym) + 1
        self.sym = SymTensors.from_list(sym_info, dev, s1=self.s1)
        self.sym = SymTensors(self.sym.cands.clone(), self.sym.valid.clone(), self.sym.is_sym.clone())  # own static buffers
        self._B = B

        def iteration():
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp):
                out, ld = self.model(self.static["x"], self.static["tfd_kps"], init_pose=self.static["init_pose"],
                                     init_scale=self.static["init_scale"], K_zoom=self.static.get("K_zoom"),
                                     gt_ego_rot=self.static["gt_ego_rot"], gt_trans=self.static["gt_trans"],
                                     gt_scale=self.static["gt_scale"], obj_kps=self.static["obj_kps"],
                                     mean_scales=self.static.get("mean_scales"), sym_info=self.sym, do_loss=True,
                                     cur_iter=1)
            loss = sum(ld.values())
            loss.backward()
            return out, ld

        # warm-up on a side stream (allocates scratch, optimizer state, the packed-weight buffers).  The steps it
        # takes are undone afterwards: parameters and optimizer state are restored to what the caller handed in.
        params = [p for g in self.opt.param_groups for p in g["params"]]
        snap_p = [p.detach().clone() for p in params]
        snap_s = {p: (st["step"], st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["slow_buffer"].clone())
                  for p, st in self.opt.state.items() if "exp_avg" in st}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.opt.zero_grad(set_to_none=True)
                iteration()
                self.opt.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for p, q in zip(params, snap_p):
                p.copy_(q)
            for p, st in self.opt.state.items():
                if "exp_avg" not in st:
                    continue
                if p in snap_s:
                    st["step"] = snap_s[p][0]
                    st["exp_avg"].copy_(snap_s[p][1]); st["exp_avg_sq"].copy_(snap_s[p][2]); st["slow_buffer"].copy_(snap_s[p][3])
                else:  # state created by the warm-up: back to a fresh optimizer's
                    st["step"] = 0
                    st["exp_avg"].zero_(); st["exp_avg_sq"].zero_(); st["slow_buffer"].copy_(p)
        hip.bump_param_epoch()
        torch.cuda.synchronize(dev)

        # capture: gradients are allocated inside the graph's private pool, so their addresses are fixed
        self.opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        # captured on the warm-up stream, so the kernels point at THAT stream's scratch buffers (train_ops._ws and the
        # runtime's workspace are per (device, stream)), which the warm-up has already grown to their final size
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="relaxed"):
            self.out, self.losses = iteration()
            # host half runs now (addresses of the captured gradients go into the record table) ...
            self.opt.prepare_step(wait=False)
            # ... and only the kernels are captured; the table upload is issued eagerly before every replay
            self.opt.launch_step(upload=False)
        # the buffers whose addresses are baked into the graph are owned here, whatever the per-stream caches do later
        # (an eager step that outgrows a cache entry replaces it; the captured block must not be recycled under the graph)
        from . import train_ops

        rt = self.model._runtime() if hasattr(self.model, "_runtime") else None
        self._keep = (train_ops.scratch_of(dev, side), side,
                      None if rt is None or rt._ws is None else rt._ws.get((dev.index, side.cuda_stream)),
                      None if rt is None else rt._packed)
        # the gradient tensors the captured backward writes and the captured optimizer step reads: re-attached before every
        # replay (an eager `zero_grad(set_to_none=True)` in between detaches them, and the host half of the step would
        # then find no gradient to describe)
        self._grads = [(p, p.grad) for p in params if p.grad is not None]
        # the packs recorded above did not execute: the next eager forward must re-pack (HipRuntime.params also refuses
        # to call a pack fresh while capturing)
        hip.bump_param_epoch()
        # the capture did not execute anything: undo the step counter it advanced
        for st in self.opt.state.values():
            if "step" in st:
                st["step"] -= 1

    @torch.no_grad()
    def __call__(self, sym_info=None, **inputs):
        for k, v in inputs.items():
            if k not in self.static:
                raise KeyError(f"unknown or uncaptured input {k!r}")
            if v.shape != self.static[k].shape:
                raise ValueError(f"{k}: captured with shape {tuple(self.static[k].shape)}, got {tuple(v.shape)}")
            self.static[k].copy_(v, non_blocking=True)
        if sym_info is not None:
            new = SymTensors.from_list(sym_info, self.sym.cands.device, s1=self.s1)
            if new.cands.shape != self.sym.cands.shape:
                raise ValueError("more symmetry candidates than the graph was captured for (max_sym)")
            self.sym.cands.copy_(new.cands, non_blocking=True)
            self.sym.valid.copy_(new.valid, non_blocking=True)
            self.sym.is_sym.copy_(new.is_sym, non_blocking=True)
        for p, g in self._grads:
            if p.grad is not g:
                p.grad = g
        self.opt.prepare_step()   # host: step counters, RAdam scalars -> pinned table
        self.opt.upload_table()   # stream-ordered before the replay
        self.graph.replay()
        hip.bump_param_epoch()
        return self.out, self.losses


_REFINE_INPUTS = ("pcl", "obj_kps", "obj_pose_est", "obj_scale_est", "K", "obj_mean_scales")


class GraphedRefine:
    """``model.refine`` for a fixed (B, N, M, K) as ONE HIP-graph replay.

    The evaluator's operating point (one image = a handful of objects per call, ``catre_evaluator.py:292-311``) is a chain
    of 14 short launches per iteration: at B=1 the K=4 loop is 0.60 ms of GPU time and 0.24 ms of host time, all of it
    ``hipLaunchKernel`` (56 launches).  One stream hides that behind the GPU; several images refined concurrently on
    separate streams do not - four streams saturate at 3.5 k refines/s because the HOST is busy launching
    (``profiles/multi_stream_probe.py``).  Replaying a captured graph costs the host one packed copy of the inputs (a
    single ``torch.cat`` into the static buffer) and one graph launch:

        g = GraphedRefine(model, example_batch)          # warm-up + capture on a side stream
        out = g(batch)                                   # pose_0..pose_K / scale_0..scale_K, same keys as model.refine

    The outputs are static tensors that the next call overwrites (``clone()`` what must survive).  Shapes and ``n_iter``
    are fixed at capture; the weights are re-checked on every call (an in-place update is re-packed before the replay, a
    re-allocated parameter triggers a new capture).  Replays are bit-identical to ``model.refine``.  One instance per
    stream for concurrent use (each owns its inputs, outputs and scratch).
    """

    def __init__(self, model, example, n_iter=None, warmup=2):
        self.model = model
        self.n_iter = int(model.cfg.MODEL.CATRE.N_ITER_TEST if n_iter is None else n_iter)
        self.keys = [k for k in _REFINE_INPUTS if example.get(k) is not None]
        for k in ("pcl", "obj_kps", "obj_pose_est", "obj_scale_est"):
            if k not in self.keys:
                raise KeyError(f"example batch lacks {k!r}")
        dev = example["pcl"].device
        self.dev = dev
        self.shapes = {k: tuple(example[k].shape) for k in self.keys}
        sizes = [int(example[k].numel()) for k in self.keys]
        self.flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        self.static, off = {}, 0
        for k, n in zip(self.keys, sizes):
            self.static[k] = self.flat[off:off + n].view(self.shapes[k])
            off += n
        self._stage(example)
        self._capture(warmup)

    def _stage(self, batch):
        parts = []
        for k in self.keys:
            v = batch[k]
            if tuple(v.shape) != self.shapes[k]:
                raise ValueError(f"{k}: captured with shape {self.shapes[k]}, got {tuple(v.shape)}")
            parts.append(hip.require_dev_f32(v, k, contiguous=False).reshape(-1))
        torch.cat(parts, out=self.flat)  # one launch for all inputs

    def _param_ptrs(self):
        return tuple(t.data_ptr() if t is not None else 0 for t in self.model._runtime()._live_params())

    def _capture(self, warmup):
        dev, rt = self.dev, self.model._runtime()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(max(1, warmup)):  # allocates this stream's workspace, packs the weights
                self.model.refine(self.static, n_iter=self.n_iter)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=side, capture_error_mode="relaxed"):
            self.out = self.model.refine(self.static, n_iter=self.n_iter)
        # the scratch buffer the captured kernels point at: owned here, whatever the runtime's per-stream cache does
        self._ws = rt._ws.get((dev.index, side.cuda_stream))
        self._packed = rt._packed
        self._ptrs = self._param_ptrs()
        self._side = side

    @torch.no_grad()
    def __call__(self, batch):
        rt = self.model._runtime()
        with torch.cuda.device(self.dev):
            rt.params(self.dev)  # stale packed weights are re-packed in place, stream-ordered before the replay
            if rt._packed is not self._packed or self._param_ptrs() != self._ptrs:
                self._capture(1)  # parameters moved: the captured pointers are stale
            self._stage(batch)
            self.graph.replay()
        return self.out
