"""Ranger (RAdam + Lookahead + gradient centralization) as ONE fused multi-tensor HIP step.

Same class name, constructor arguments, defaults and per-parameter ``state`` keys (``step``, ``exp_avg``,
``exp_avg_sq``, ``slow_buffer``) as the reference's ``lib/torch_utils/solver/ranger.py:31-202`` - so optimizer
checkpoints written by either load into the other - but the update of all ~70 parameter tensors is two kernel
launches (``catre_op_ranger_step``) instead of ~1000 tiny torch ops per step, K times per data batch.  The
train loop's ``nan_to_num(grad, nan=0, posinf=1e5, neginf=-1e5)`` (``core/catre/engine/engine.py:351-353``) can
be folded into the same pass with ``clean_grads=True``.

The scalar RAdam rectification terms (N_sma, step size) are computed on the host in double precision exactly
like the reference; everything per element runs on the device in fp32.  ``step()`` is split in two halves so that a
captured HIP graph can replay it: :meth:`prepare_step` (host only: advances the step counters and fills a pinned
table of pointers and scalars) and :meth:`launch_step` (device only: one asynchronous copy of that table and the
kernels).
"""
import math

import numpy as np
import torch
from torch.optim.optimizer import Optimizer

from . import hip

_CHUNK = 4096
_REC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("slow", "<u8"), ("numel", "<i4"),
                 ("row_len", "<i4"), ("row_off", "<i4"), ("lr_step", "<f4"), ("wd_lr", "<f4"), ("adaptive", "<i4"),
                 ("lookahead", "<i4"), ("pad", "<i4")])
assert _REC.itemsize == 72


class Ranger(Optimizer):
    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(0.95, 0.999), eps=1e-5,
                 weight_decay=0, use_gc=True, gc_conv_only=False, clean_grads=False, grad_limit=1e5):
        if not 0.0 <= alpha <= 1.0:
            raise ValueError(f"Invalid slow update rate: {alpha}")
        if not 1 <= k:
            raise ValueError(f"Invalid lookahead steps: {k}")
        if not lr > 0:
            raise ValueError(f"Invalid Learning Rate: {lr}")
        if not eps > 0:
            raise ValueError(f"Invalid eps: {eps}")
        defaults = dict(lr=lr, alpha=alpha, k=k, step_counter=0, betas=betas, N_sma_threshhold=N_sma_threshhold, eps=eps,
                        weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.N_sma_threshhold = N_sma_threshhold
        self.alpha = alpha
        self.k = k
        self.use_gc = use_gc
        self.gc_gradient_threshold = 3 if gc_conv_only else 1
        self.clean_grads = bool(clean_grads)
        self.grad_limit = float(grad_limit)
        self._layout = None   # cached per set of (param, grad) addresses: chunk / row tables, pinned + device record table
        self._pending = None  # what prepare_step hands to launch_step

    # ---------------------------------------------------------------- scalar RAdam terms (reference :150-170)
    def _step_terms(self, step, beta1, beta2):
        beta2_t = beta2 ** step
        n_max = 2 / (1 - beta2) - 1
        n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
        if n_sma > self.N_sma_threshhold:
            size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) / (
                1 - beta1 ** step)
            return size, True
        return 1.0 / (1 - beta1 ** step), False

    def prepare_step(self, wait=True):
        """Host half of a step: state initialisation, step counters, RAdam scalars -> the pinned record table.
        Returns False when no parameter has a gradient.  wait=False skips the (host-blocking) wait for the slot's
        previous upload - only for callers that know it has completed, e.g. during graph capture."""
        groups_betas = {tuple(g["betas"]) for g in self.param_groups}
        groups_eps = {g["eps"] for g in self.param_groups}
        if len(groups_betas) != 1 or len(groups_eps) != 1:
            raise NotImplementedError("fused Ranger: betas / eps must be the same in every param group (lr, weight_decay, k may differ)")
        (beta1, beta2), eps = next(iter(groups_betas)), next(iter(groups_eps))
        # One pass over the parameters fills plain lists; the pinned record table is then written column by column (a
        # structured-array row assignment per tensor, two checker calls and a fresh rectification term per tensor were
        # 0.65 ms of a 4 ms host iteration; this is 0.15).
        f32 = torch.float32
        ps, gs, sts = [], [], []
        c_lr, c_wd, c_ad, c_look, c_rl, c_n = [], [], [], [], [], []
        terms = {}
        dev = None
        for group in self.param_groups:
            lr, wd, k = group["lr"], group["weight_decay"], group["k"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("Ranger optimizer does not support sparse gradients")
                if not (p.is_cuda and p.dtype is f32):
                    hip.require_dev_f32(p, "parameter")   # raises with the library's message
                if not g.is_contiguous():
                    g = g.contiguous()
                if not (g.is_cuda and g.dtype is f32):
                    hip.require_dev_f32(g, "gradient")
                if not p.is_contiguous():
                    raise ValueError("fused Ranger needs contiguous parameters")
                dev = p.device
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["slow_buffer"] = p.detach().clone(memory_format=torch.contiguous_format)
                step = st["step"] + 1
                st["step"] = step
                t = terms.get(step)
                if t is None:
                    t = terms[step] = self._step_terms(step, beta1, beta2)
                gc = self.use_gc and p.dim() > self.gc_gradient_threshold
                n = p.numel()
                ps.append(p), gs.append(g), sts.append(st)
                c_lr.append(t[0] * lr), c_wd.append(wd * lr), c_ad.append(int(t[1])), c_look.append(int(step % k == 0))
                c_n.append(n), c_rl.append((n // p.shape[0]) if gc else 0)
        if not ps:
            self._pending = None
            return False
        nrec = len(ps)
        p_ptr = [p.data_ptr() for p in ps]
        key = tuple(zip(p_ptr, c_n, c_rl))
        if self._layout is None or self._layout["key"] != key:
            chunks, row_tensor = [], []
            for ti in range(nrec):
                n = c_n[ti]
                chunks += [(ti, o) for o in range(0, n, _CHUNK)]
                if c_rl[ti] > 0:
                    row_tensor += [ti] * (n // c_rl[ti])
            # two pinned slots, used alternately: the host may fill one while the asynchronous upload of the other
            # is still queued behind earlier GPU work
            hosts = [torch.empty(nrec * _REC.itemsize, dtype=torch.uint8).pin_memory() for _ in range(2)]
            row_off, off = [], 0
            for ti in range(nrec):
                row_off.append(off)
                if c_rl[ti] > 0:
                    off += c_n[ti] // c_rl[ti]
            self._layout = dict(
                key=key, n_chunks=len(chunks), total_rows=len(row_tensor), row_off=row_off,
                chunks=torch.tensor(chunks, dtype=torch.int32).to(dev),
                rows=torch.tensor(row_tensor if row_tensor else [0], dtype=torch.int32).to(dev),
                ws=torch.empty(max(len(row_tensor), 1), dtype=torch.float32, device=dev),
                hosts=hosts, tables=[h.numpy().view(_REC) for h in hosts], events=[None, None], slot=0,
                dev=torch.empty(nrec * _REC.itemsize, dtype=torch.uint8, device=dev),
            )
        L = self._layout
        L["slot"] ^= 1
        if wait and L["events"][L["slot"]] is not None:
            L["events"][L["slot"]].synchronize()  # its previous upload (two steps ago) has long been consumed
        table = L["tables"][L["slot"]]
        table["p"] = p_ptr
        table["g"] = [g.data_ptr() for g in gs]
        table["m"] = [st["exp_avg"].data_ptr() for st in sts]
        table["v"] = [st["exp_avg_sq"].data_ptr() for st in sts]
        table["slow"] = [st["slow_buffer"].data_ptr() for st in sts]
        table["numel"], table["row_len"], table["row_off"] = c_n, c_rl, L["row_off"]
        table["lr_step"], table["wd_lr"], table["adaptive"], table["lookahead"], table["pad"] = c_lr, c_wd, c_ad, c_look, 0
        # the (possibly copied-to-contiguous) gradients must outlive the launch
        self._pending = dict(n=nrec, beta1=beta1, beta2=beta2, eps=eps, device=dev, keep=gs)
        return True

    def upload_table(self):
        """Asynchronous copy of the record table prepare_step filled (current stream).  Not part of a captured graph:
        a replaying caller issues it before every replay."""
        pend, L = self._pending, self._layout
        if pend is None:
            return
        with torch.cuda.device(pend["device"]):
            L["dev"].copy_(L["hosts"][L["slot"]], non_blocking=True)
            ev = L["events"][L["slot"]] or torch.cuda.Event()
            ev.record()
            L["events"][L["slot"]] = ev

    def launch_step(self, upload=True):
        """Device half: the fused kernels on the current stream (after the table upload unless the caller did it).
        With upload=False this is capturable: only kernel launches on fixed device addresses."""
        pend, L = self._pending, self._layout
        if pend is None:
            return
        if upload:
            self.upload_table()
        lib = hip.load()
        dev = pend["device"]
        with torch.cuda.device(dev):
            hip.check(lib.catre_op_ranger_step(hip.ptr(L["dev"]), pend["n"], hip.ptr(L["chunks"]), L["n_chunks"],
                                               hip.ptr(L["rows"]), L["total_rows"], hip.ptr(L["ws"]), pend["beta1"],
                                               pend["beta2"], pend["eps"], self.alpha, int(self.clean_grads),
                                               self.grad_limit, hip.stream_ptr(dev)), "catre_op_ranger_step")
        hip.bump_param_epoch()  # the kernel wrote the parameters behind torch's back: invalidate packed-weight caches

    @torch.no_grad()
    def step(self, closure=None):
        if self.prepare_step():
            self.launch_step()
        return None
