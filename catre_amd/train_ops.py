"""Training ops: ``torch.autograd.Function`` wrappers whose forward AND backward are HIP kernels
(``catre_op_*`` in ``include/catre_hip.h``).  torch supplies tensors, the autograd graph and pure data
movement (``cat`` / ``pad`` / ``reshape`` / ``contiguous``); no torch arithmetic op touches activations.

Row orders used below: "cloud-major" = B*N observed rows then B*M prior rows; "object-major" =
[N observed | M prior] per object.
"""
import ctypes
import logging
import os
import threading
import typing

import torch
import torch.nn.functional as F

from . import hip

logger = logging.getLogger(__name__)
_scratch = {}


def _ws(nbytes, device):
    """Grow-only scratch, one buffer per (device, stream) like ``HipRuntime.workspace``: every op that uses it runs on the
    current stream, so reuse on one stream is ordered, and training iterations on different streams (or a captured
    ``GraphedTrainStep`` next to eager steps) never share split-K partials.  A buffer that is outgrown is dropped here
    only - whoever captured its address (``GraphedTrainStep``) keeps its own reference (``scratch_of``)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), hip.stream_ptr(device).value or 0)
    buf = _scratch.pop(key, None)
    if buf is None or buf.numel() < nbytes:
        if buf is None and len(_scratch) >= 32:
            # least recently used stream only (every use re-inserts its key at the end): the caching allocator keeps a
            # block alive until its queued work is done; a capture that baked an address in keeps its own reference
            old = next(iter(_scratch))
            del _scratch[old]
            logger.info("train_ops: scratch of stream %#x evicted (32 streams cached)", old[1])
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
    _scratch[key] = buf
    return buf


def scratch_of(device, stream):
    """The scratch buffer training ops issued on ``stream`` currently use (None if none yet)."""
    return _scratch.get((device.index if device.index is not None else torch.cuda.current_device(), stream.cuda_stream))


def _st(t):
    return hip.stream_ptr(t.device)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _pad_cols(t, mult):
    """t [..., k] with zero columns appended up to a multiple of `mult` (data movement only; called on tensors outside the
    autograd graph - inside Function.forward / backward).  2-D fp32 device tensors: one launch (catre_op_pad_cols), any
    strides; everything else through F.pad."""
    k = t.shape[-1]
    r = (-k) % mult
    if r == 0:
        return t
    if (t.dim() == 2 and t.is_cuda and t.dtype == torch.float32 and t.numel() > 0
            and (not t.requires_grad or not torch.is_grad_enabled())):  # (grad mode is off inside Function.forward / backward)
        out = torch.empty(t.shape[0], k + r, dtype=torch.float32, device=t.device)
        hip.check(hip.load().catre_op_pad_cols(hip.ptr(t), t.stride(0), t.stride(1), t.shape[0], k, hip.ptr(out), k + r,
                                               _st(t)), "catre_op_pad_cols")
        return out
    return F.pad(t, (0, r))


# ------------------------------------------------------------------------------------------------- GEMMs
# Mixed precision: inside torch.autocast (what SOLVER.AMP.ENABLED does to the reference's forward, engine.py:304) the
# forward, dgrad and wgrad GEMMs run with bf16 operands and fp32 accumulation / outputs (bias gradients stay fp32
# sums).  `amp_mode("fp32" | "bf16" | "split")` overrides the autocast state (cfg.MODEL.CATRE.COMPUTE_DTYPE); "split"
# keeps fp32-grade results on the bf16 pipe (hi + lo bf16 operands, three products - DESIGN 5e).
# The mode is the C ABI's compute_dtype: 0 = CATRE_DTYPE_F32, 1 = CATRE_DTYPE_BF16, 2 = CATRE_DTYPE_SPLIT.
_TLS = threading.local()   # per-thread: the compute mode override and the kernel-selection knobs of the forward in progress
_MODES = {"fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "split": 2}


class amp_mode:
    def __init__(self, mode):
        assert mode is None or mode in _MODES, mode
        self.mode = None if mode is None else _MODES[mode]

    def __enter__(self):
        self.prev = getattr(_TLS, "amp", None)
        _TLS.amp = self.mode if self.mode is not None else self.prev

    def __exit__(self, *a):
        _TLS.amp = self.prev


_warned_fp16 = [False]


def autocast_on():
    """``torch.is_autocast_enabled()`` - and, once per process, a warning when the request is for fp16 (what the reference's
    ``autocast(enabled=AMP_ON)`` asks for on "cuda", engine.py:304): the reduced-precision kernels of this library take
    bf16 operands (8-bit significands, fp32 accumulation and outputs, no loss scaling needed), whatever dtype autocast names."""
    if not torch.is_autocast_enabled():
        return False
    if not _warned_fp16[0]:
        try:
            dt = torch.get_autocast_dtype("cuda")
        except Exception:  # older torch
            dt = torch.get_autocast_gpu_dtype()
        if dt == torch.float16:
            _warned_fp16[0] = True
            logger.warning("torch.autocast asks for float16: catre_amd's reduced-precision kernels use BFLOAT16 operands "
                           "(fp32 accumulation / outputs) for any autocast dtype - 8-bit significands instead of 11.  "
                           "MODEL.CATRE.COMPUTE_DTYPE='split' keeps fp32-grade results on the bf16 matrix pipe, 'fp32' "
                           "ignores autocast.")
    return True


def _amp():
    ov = getattr(_TLS, "amp", None)
    return int(autocast_on()) if ov is None else ov


class TrainKernels(typing.NamedTuple):
    """Which kernels the training forward picks where more than one form exists (A/B measurements, tests).  Per MODEL:
    ``cfg.MODEL.CATRE.TRAIN_KERNELS = dict(fused_lp_rot=False, ...)`` is applied around that model's forward; the
    environment variables only set the process defaults."""
    split_l0_sp: bool = os.environ.get("CATRE_SPLIT_L0_KERNEL", "split") != "fp32"      # split mode's first block on k_rot_l0_bwd_sp (else fp32 k_rot_l0_bwd)
    lp_rot_fuse_gn0: bool = os.environ.get("CATRE_LP_ROT_GN0", "fused") != "separate"    # autocast head: GroupNorm-0 + GELU inside the second linear's staging
    lp_rot_bf16_rows: bool = os.environ.get("CATRE_LP_ROT_ROWS", "bf16") != "fp32"       # autocast heads' [rows,256] activations as bf16 rows (_RotHeadLP)
    fused_lp_rot: bool = os.environ.get("CATRE_LP_ROT_FUSED", "1") != "0"                # autocast rot heads as fused nodes (else layer-wise)
    split_l0_one_pass: bool = os.environ.get("CATRE_SPLIT_L0", "onepass") != "layerwise"  # split mode: first rot-head block as one node
    split_l1_one_pass: bool = os.environ.get("CATRE_SPLIT_L1", "onepass") != "layerwise"  # split mode: second block + tail as one node
    stn_recompute: bool = os.environ.get("CATRE_STN_RECOMPUTE", "1") != "0"              # fp32: STN stacks recompute their activation rows in the backward instead of saving them
    fc_bwd_one_launch: bool = os.environ.get("CATRE_FC_BWD", "1") != "0"                 # linear layers on < 2048 rows: dgrad + wgrad + bias gradient in one launch (catre_op_fc_bwd)


_DEFAULT_KNOBS = TrainKernels()


def knobs():
    return getattr(_TLS, "knobs", None) or _DEFAULT_KNOBS


class train_kernels:
    """``with train_kernels(fused_lp_rot=False):`` - knob overrides for the training forwards issued inside (this thread)."""

    def __init__(self, overrides=None, **kw):
        self.ov = dict(overrides or {}, **kw)
        unknown = set(self.ov) - set(TrainKernels._fields)
        if unknown:
            raise ValueError(f"MODEL.CATRE.TRAIN_KERNELS: unknown keys {sorted(unknown)} (known: {TrainKernels._fields})")

    def __enter__(self):
        self.prev = getattr(_TLS, "knobs", None)
        _TLS.knobs = knobs()._replace(**{k: bool(v) for k, v in self.ov.items()})

    def __exit__(self, *a):
        _TLS.knobs = self.prev


def _pack_bf16(w, J, K, dev, transpose=0):
    """w [J,K] (transpose: w is the [K,J] source of the [J,K] weight) -> bf16 fragments."""
    wp = torch.empty(J * K // 2, dtype=torch.float32, device=dev)  # J*K bf16
    hip.check(hip.load().catre_op_pack_bf16(hip.ptr(w), w.stride(0), J, K, int(transpose), hip.ptr(wp), _st(w)),
              "catre_op_pack_bf16")
    return wp


def _pack_split(w, J, K, dev, transpose=0):
    wp = torch.empty(J * K, dtype=torch.float32, device=dev)  # 2*J*K bf16: hi pack, lo pack
    hip.check(hip.load().catre_op_pack_split(hip.ptr(w), w.stride(0), J, K, int(transpose), hip.ptr(wp), _st(w)),
              "catre_op_pack_split")
    return wp


def _tiled_gemm_ok(R, J, K, min_rows=2048):
    """Shapes the tiled row kernel (catre_op_gemm_rows) takes."""
    # (64-row tiles: under ~2048 rows - the FC tails, whose rows are clouds - the tiled kernel would put a handful of
    # workgroups on 256 CUs; the split-K catre_linear takes those)
    return (J % 32 == 0) and (J <= 256 or J in (512, 1024)) and (K in (8, 16, 32, 64, 128) or K % 256 == 0) and R >= min_rows


def _gemm_nt(x, w, bias, relu, mask=None, identity_k=0, xmask=None, amp=False, wT=None):
    """y[R,J] = act(x[R,K] w[J,K]^T + bias) (zeroed where mask<=0); with xmask the left operand is x .* (xmask > 0).
    Picks the tiled row kernel when the shape allows, else the one-block-per-32x32 kernel (small R or odd J).
    wT (instead of w): the weight is the transpose of this contiguous [K,J] matrix (the dgrad of a linear: x = dy,
    wT = the layer's own weight) - the fragment pack reads it transposed, no copy."""
    lib = hip.load()
    R, K = x.shape
    tr = 0
    if wT is not None:
        assert w is None and wT.shape[0] == K
        J = wT.shape[1]
    else:
        J = w.shape[0]
    dev = x.device
    y = torch.empty(R, J, dtype=torch.float32, device=dev)
    big = _tiled_gemm_ok(R, J, K) and identity_k == 0
    if wT is not None:
        if big:
            w, tr = wT, 1
        elif K % 8 == 0 and bias is None and not relu and mask is None and identity_k == 0 \
                and (xmask is None or xmask.stride(0) == x.stride(0)):
            # small dgrad: the kernel reads the layer's own [K_contraction][J] weight (no transposed copy) and applies the
            # ReLU mask of the layer's output while it loads dy
            hip.check(lib.catre_linear_t(hip.ptr(x), x.stride(0), hip.ptr(xmask), hip.ptr(wT), wT.stride(0), hip.ptr(y), J, R,
                                         J, K, _st(x)), "catre_linear_t")
            return y
        else:
            w = _c(wT.t())
    if big and amp in (1, 2) and K in (64, 128, 256, 512):
        wp = (_pack_bf16 if amp == 1 else _pack_split)(w, J, K, dev, tr)
        fn = lib.catre_op_gemm_rows_bf16 if amp == 1 else lib.catre_op_gemm_rows_split
        hip.check(fn(hip.ptr(x), x.stride(0), hip.ptr(xmask), xmask.stride(0) if xmask is not None else 0, hip.ptr(wp),
                     hip.ptr(bias), hip.ptr(mask), mask.stride(0) if mask is not None else 0, hip.ptr(y), J, R, J, K,
                     int(relu), _st(x)), "catre_op_gemm_rows_bf16" if amp == 1 else "catre_op_gemm_rows_split")
    elif big:
        wp = torch.empty(J * K, dtype=torch.float32, device=dev)
        hip.check(lib.catre_op_pack(hip.ptr(w), w.stride(0), J, K, tr, hip.ptr(wp), _st(x)), "catre_op_pack")
        hip.check(lib.catre_op_gemm_rows_m(hip.ptr(x), x.stride(0), hip.ptr(xmask),
                                           xmask.stride(0) if xmask is not None else 0, hip.ptr(wp), hip.ptr(bias),
                                           hip.ptr(mask), mask.stride(0) if mask is not None else 0, hip.ptr(y), J, R, J,
                                           K, int(relu), _st(x)), "catre_op_gemm_rows_m")
    else:
        assert K % 8 == 0 and mask is None and xmask is None
        hip.check(lib.catre_linear(hip.ptr(x), x.stride(0), hip.ptr(w), w.stride(0), hip.ptr(bias), hip.ptr(y), J, R, J,
                                   K, int(relu), int(identity_k), _st(x)), "catre_linear")
    return y


def _gemm_tn(dy, x, with_bias=False, ymask=None, amp=0):
    """dW[J,K] = dy[R,J]^T x[R,K] (deterministic split reduction); with_bias also returns db[J] = column sums of dy,
    taken from the tiles the kernel stages anyway; with ymask, dy .* (ymask > 0) replaces dy (ReLU backward)."""
    lib = hip.load()
    R, J = dy.shape
    K = x.shape[1]
    dy4, x4 = _pad_cols(dy, 4), _pad_cols(x, 4)
    ym4 = _pad_cols(ymask, 4) if ymask is not None else None
    J4, K4 = dy4.shape[1], x4.shape[1]
    buf = torch.empty(J4 * K4 + (J4 if with_bias else 0), dtype=torch.float32, device=dy.device)
    dw = buf[: J4 * K4].view(J4, K4)
    db = buf[J4 * K4:] if with_bias else None  # right behind dW: one split reduction covers both
    need = lib.catre_op_gemm_tn_bias_ws_bytes(J4, K4, R)
    ws = _ws(need, dy.device)
    hip.check(lib.catre_op_gemm_tn_bias_lp(hip.ptr(dy4), dy4.stride(0), hip.ptr(ym4),
                                           ym4.stride(0) if ym4 is not None else 0, hip.ptr(x4), x4.stride(0), hip.ptr(dw),
                                           hip.ptr(db), J4, K4, R, 0, hip.ptr(ws), ws.numel(), int(amp), _st(dy)),
              "catre_op_gemm_tn_bias_lp")
    if with_bias:
        return dw[:J, :K], db[:J]
    return dw[:J, :K]


def _wsum_backward(dout, y3, wv, B, P, has_bp, has_bn):
    """conv_p backward behind the 256 -> 3 neck: dy3[b,p,:] = wp[p] dout[b,:], dwp, and the two bias gradients (conv_p's
    and - has_bn - the neck's, the column sums of dy3) -> (dy3, dwp, dbp, dbn)."""
    dev = dout.device
    dy3 = torch.empty(B * P, 3, dtype=torch.float32, device=dev)
    dwp = torch.empty(P, dtype=torch.float32, device=dev)
    dbp = torch.empty(1, dtype=torch.float32, device=dev) if has_bp else None
    dbn = torch.empty(3, dtype=torch.float32, device=dev) if has_bn else None
    ws = _ws(B * P * 4, dev)
    hip.check(hip.load().catre_op_wsum_bwd_n(hip.ptr(dout), hip.ptr(y3), hip.ptr(wv), hip.ptr(dy3), hip.ptr(dwp), hip.ptr(dbp),
                                             hip.ptr(dbn), 0, hip.ptr(ws), ws.numel(), B, P, _st(dout)), "catre_op_wsum_bwd_n")
    return dy3, dwp, dbp, dbn


def _colsum(dy):
    lib = hip.load()
    R, J = dy.shape
    out = torch.empty(J, dtype=torch.float32, device=dy.device)
    ws = _ws(256 * J * 4, dy.device)
    hip.check(lib.catre_op_colsum(hip.ptr(dy), dy.stride(0), R, J, hip.ptr(out), 0, hip.ptr(ws), ws.numel(), _st(dy)),
              "catre_op_colsum")
    return out


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b); W may be a Conv1d weight [J,K,1].  identity_k adds vec(I_k) (STN tails)."""

    @staticmethod
    def forward(ctx, x, w, b, relu, identity_k, pre):
        w2 = w.reshape(w.shape[0], -1)
        K = w2.shape[1]
        amp = _amp()
        if pre is not None:
            # the output already exists (or is about to be written on this stream by a fused forward kernel,
            # catre_train_*_fwd): this node only ties it into the graph; the backward below is the layer's own
            y = pre
        else:
            # (a weight that is a column slice of a wider matrix - rot-head layer 0's global half - is read in place: every
            # kernel below takes its leading dimension)
            xk, wk = _c(_pad_cols(x, 8)), _pad_cols(w2, 8)
            if wk.stride(1) != 1 or wk.stride(0) % 4:
                wk = _c(wk)
            y = _gemm_nt(xk, wk, b, relu, identity_k=identity_k, amp=amp)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.relu, ctx.K, ctx.has_b, ctx.amp = relu, K, b is not None, amp
        ctx.fc1 = knobs().fc_bwd_one_launch
        return y

    @staticmethod
    def backward(ctx, dy):
        return _linear_backward(ctx, dy)


def _skinny_bwd_ok(dy, x, w2):
    """Shapes catre_op_skinny_bwd takes: 64 output channels, <= 4 live input columns in rows of 4 or 8 floats."""
    return dy.shape[1] == 64 and w2.shape[0] == 64 and w2.shape[1] <= 4 and x.shape[1] in (4, 8) and dy.shape[0] > 0


def _skinny_backward(dy, dy2, ymask, x, w2, need_dx):
    lib = hip.load()
    R = dy.shape[0]
    dev = dy.device
    Kw = w2.shape[1]
    buf = torch.empty(64 * 4 + 64, dtype=torch.float32, device=dev)
    dx = torch.empty(R, x.shape[1], dtype=torch.float32, device=dev) if need_dx else None
    wc = _c(w2)
    ws = _ws(lib.catre_op_gemm_tn_bias_ws_bytes(64, 4, R), dev)
    hip.check(lib.catre_op_skinny_bwd(hip.ptr(dy), dy.stride(0), hip.ptr(dy2), dy2.stride(0) if dy2 is not None else 0,
                                      hip.ptr(ymask), ymask.stride(0) if ymask is not None else 0, hip.ptr(x), x.stride(0),
                                      hip.ptr(wc), wc.stride(0), Kw, hip.ptr(buf), hip.ptr(buf[256:]), hip.ptr(dx),
                                      x.shape[1] if need_dx else 0, x.shape[1], R, hip.ptr(ws), ws.numel(), _st(dy)),
              "catre_op_skinny_bwd")
    return dx, _c(buf[:256].view(64, 4)[:, :Kw]), buf[256:]


def _fc_backward(ctx, dy, x, w, w2, y):
    """The layer's whole backward in one launch (rows are clouds or objects: the FC tails, the ts head, the rot heads'
    global halves): ReLU mask, dgrad, wgrad and bias gradient, any widths - no padded or transposed copies."""
    R, J = dy.shape
    Kx, Kw = x.shape[1], w2.shape[1]
    dev = dy.device
    need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    need_db = ctx.has_b and ctx.needs_input_grad[2]
    ym = None
    if ctx.relu:
        ym = y if (y.stride(1) == 1 and y.stride(0) == dy.stride(0)) else None
        if ym is None:
            ym, dy = _c(y), _c(dy)
    if x.stride(1) != 1:
        x = _c(x)
    if w2.stride(1) != 1:
        w2 = _c(w2)
    dx = torch.empty(R, Kx, dtype=torch.float32, device=dev) if need_dx else None
    # (the bias gradient comes out of the weight gradient's workgroups: a frozen weight still runs them)
    buf = torch.empty(J * Kw + J, dtype=torch.float32, device=dev) if (need_dw or need_db) else None
    dw = buf[: J * Kw].view(J, Kw) if buf is not None else None
    db = buf[J * Kw:] if need_db else None
    hip.check(hip.load().catre_op_fc_bwd(hip.ptr(dy), dy.stride(0), hip.ptr(ym), hip.ptr(x), x.stride(0), hip.ptr(w2),
                                         w2.stride(0), hip.ptr(dx), hip.ptr(dw), hip.ptr(db), R, J, Kx, Kw, int(ctx.amp),
                                         _st(dy)), "catre_op_fc_bwd")
    return dx, dw.reshape(w.shape) if need_dw else None, db, None, None, None


def _linear_backward(ctx, dy, dy2=None):
    x, w, y = ctx.saved_tensors
    lib = hip.load()
    w2 = w.reshape(w.shape[0], -1)
    if dy is None:
        dy, dy2 = dy2, None
    dy = _c(dy)
    if _skinny_bwd_ok(dy, x, w2):
        # conv1 on 3-d points: add of the two gradient streams, ReLU mask, dgrad, wgrad and bias gradient in one pass
        dx, dw, db = _skinny_backward(dy, _c(dy2) if dy2 is not None else None, _c(y) if ctx.relu else None, _c(x), w2,
                                      ctx.needs_input_grad[0])
        return (dx, dw.reshape(w.shape) if ctx.needs_input_grad[1] else None,
                db if ctx.has_b and ctx.needs_input_grad[2] else None, None, None, None)
    if dy2 is not None:
        dy = dy + dy2
    if getattr(ctx, "fc1", False) and 0 < dy.shape[0] < 2048 and x.dim() == 2 and x.shape[0] == dy.shape[0] \
            and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
        return _fc_backward(ctx, dy, x, w, w2, y)
    ymask = None
    if ctx.relu:
        # ReLU backward: folded into the operand loads of the two GEMMs below when both take the tiled kernels,
        # else as its own pass
        R, J = dy.shape
        # (a small dgrad - under 2048 rows - takes catre_linear_t, which masks its operand as well)
        if J % 8 == 0 and (not ctx.needs_input_grad[0] or _tiled_gemm_ok(R, w2.shape[1], J)
                           or (not _tiled_gemm_ok(R, w2.shape[1], J) and w2.is_contiguous() and y.is_contiguous()
                               and dy.stride(0) == y.stride(0))):
            ymask = _c(y)
        else:
            g = torch.empty_like(dy)
            hip.check(lib.catre_op_relu_bwd(hip.ptr(dy), hip.ptr(y), hip.ptr(g), dy.numel(), _st(dy)),
                      "catre_op_relu_bwd")
            dy = g
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        # dx[R,K] = dy[R,J] W[J,K]  ==  gemm_nt(dy, W^T[K,J]); the contraction length J is padded to 8
        if dy.shape[1] % 8 == 0:
            dx = _gemm_nt(dy, None, None, False, xmask=ymask, amp=ctx.amp,
                          wT=w2 if (w2.stride(1) == 1 and w2.stride(0) % 4 == 0) else _c(w2))     # [R, K]
        else:
            wt = _c(_pad_cols(w2.t(), 8))      # [K, J8]
            dx = _gemm_nt(_c(_pad_cols(dy, 8)), wt, None, False, xmask=ymask, amp=ctx.amp)
        if x.shape[1] > dx.shape[1]:           # x carried zero padding columns beyond K
            dx = F.pad(dx, (0, x.shape[1] - dx.shape[1]))
        elif x.shape[1] < dx.shape[1]:
            dx = _c(dx[:, : x.shape[1]])
    want_db = ctx.has_b and ctx.needs_input_grad[2]
    if ctx.needs_input_grad[1]:
        kw = min(w2.shape[1], x.shape[1])
        if want_db:
            dw, db = _gemm_tn(dy, _c(x), with_bias=True, ymask=ymask, amp=ctx.amp)
            db = _c(db)
        else:
            dw = _gemm_tn(dy, _c(x), ymask=ymask, amp=ctx.amp)
        dw = dw[:, :kw]
        if kw < w2.shape[1]:
            dw = F.pad(dw, (0, w2.shape[1] - kw))
        dw = _c(dw).reshape(w.shape)
    elif want_db:
        if ymask is not None:
            dy = dy * (ymask > 0)
        db = _colsum(dy)
    return dx, dw, db, None, None, None


class _LinearFan2(torch.autograd.Function):
    """:class:`_Linear` whose output feeds TWO consumers (h1 = relu(conv1(x1)): the STNkd stack and the feature transform,
    pointnet.py:103-109): returned as two tensors on one buffer, so that the two gradient streams arrive separately and the
    backward adds them inside its own first pass instead of autograd running an add over [rows,64] before it."""

    @staticmethod
    def forward(ctx, x, w, b, relu, pre):
        ctx.set_materialize_grads(False)
        y = _Linear.forward(ctx, x, w, b, relu, 0, pre)
        return y, y.detach()

    @staticmethod
    def backward(ctx, dy_a, dy_b):
        if dy_a is None and dy_b is None:
            return None, None, None, None, None
        return _linear_backward(ctx, dy_a, dy_b)[:5]


def linear_fan2(x, w, b=None, relu=False, pre=None):
    return _LinearFan2.apply(x, w, b, relu, pre)


def linear(x, w, b=None, relu=False, identity_k=0, pre=None):
    return _Linear.apply(x, w, b, relu, identity_k, pre)


class _SplitCols(torch.autograd.Function):
    """w [J, K] -> (w[:, :k0] as a view, w[:, k0:] contiguous) (rot-head layer 0: the global and the point half of its
    1088-wide weight, conv_out_per_rot_head.py:126).  Backward: ONE cat of the two gradients - autograd's own slice backward
    is a zero-fill + copy per half and an add over [J, K]."""

    @staticmethod
    def forward(ctx, w, k0):
        ctx.k0, ctx.K = k0, w.shape[1]
        # the wide half stays a VIEW (the linear that consumes it reads it through its leading dimension: no 1 MB copy per
        # head and iteration); the narrow half is packed by kernels that want it contiguous
        return w[:, :k0], w[:, k0:].contiguous()

    @staticmethod
    def backward(ctx, da, db):
        if da is None and db is None:
            return None, None
        ref = da if da is not None else db
        if da is None:
            da = ref.new_zeros(ref.shape[0], ctx.k0)
        if db is None:
            db = ref.new_zeros(ref.shape[0], ctx.K - ctx.k0)
        return torch.cat([da, db], 1), None


class _Hub(torch.autograd.Function):
    """t [R,K] -> (t[:rc], t, t): the consumers of a tensor one of which reads only its first rc rows (the pooled feature:
    `[:B]` of it to the ts head, all of it to both rotation heads, CATRE_disR_shared.py:69,86).  Backward: ONE launch sums
    what arrived (catre_op_sum_rows) - autograd's route is a zero-fill + copy for the slice and an add per further consumer."""

    @staticmethod
    def forward(ctx, t, rc):
        ctx.set_materialize_grads(False)
        ctx.shape, ctx.rc = tuple(t.shape), rc
        return t[:rc], t.view_as(t), t.view_as(t)

    @staticmethod
    def backward(ctx, dc, da, db):
        if dc is None and da is None and db is None:
            return None, None
        ref = next(d for d in (da, db, dc) if d is not None)
        R, K = ctx.shape
        out = torch.empty(R, K, dtype=torch.float32, device=ref.device)
        da, db = (_c(d) if d is not None else None for d in (da, db))
        if dc is not None and dc.stride(1) != 1:    # (a column slice of a wider gradient is read in place: any row pitch)
            dc = _c(dc)
        hip.check(hip.load().catre_op_sum_rows(hip.ptr(da), hip.ptr(db), hip.ptr(dc), dc.stride(0) if dc is not None else 0,
                                               hip.ptr(out), R, ctx.rc, K, _st(ref)), "catre_op_sum_rows")
        return out, None


def hub(t, rc):
    """-> (t[:rc], t, t) whose gradients are summed by one launch (see :class:`_Hub`)."""
    return _Hub.apply(t, rc)


def split_cols(w, k0):
    return _SplitCols.apply(w, k0)


# ------------------------------------------------------------------------------------------------- linear + max-pool
class _LinearMaxPool(torch.autograd.Function):
    """g[C,J] = act(max over the points of each cloud of (x W^T + b)); rows cloud-major.  The [rows,J]
    pre-pool activation only lives inside forward; backward is sparse (gather / scatter at the argmax rows)."""

    @staticmethod
    def forward(ctx, x, w, b, relu, B, N, M, pre):
        lib = hip.load()
        w2 = _c(w.reshape(w.shape[0], -1))
        J, K = w2.shape
        C = 2 * B if M > 0 else B
        if pre is not None:  # (pooled maxima with the bias added, arg-max rows) from a fused forward kernel
            g, idx = pre
        else:
            g = torch.empty(C, J, dtype=torch.float32, device=x.device)
            idx = torch.empty(C, J, dtype=torch.int32, device=x.device)
        xc = _c(x)
        fused = (N % 64 == 0 and M % 64 == 0 and J % 32 == 0 and (J <= 256 or J in (512, 1024))
                 and (K in (64, 128) or K % 256 == 0))
        amp = _amp()
        if pre is not None:
            pass
        elif fused and amp in (1, 2) and K in (64, 128, 256, 512):
            wp = (_pack_bf16 if amp == 1 else _pack_split)(w2, J, K, x.device)
            need = lib.catre_op_linear_maxpool_ws_bytes(xc.shape[0], J)
            ws = _ws(need, x.device)
            fn = lib.catre_op_linear_maxpool_bf16 if amp == 1 else lib.catre_op_linear_maxpool_split
            hip.check(fn(hip.ptr(xc), xc.stride(0), hip.ptr(wp), hip.ptr(b), hip.ptr(g), hip.ptr(idx), J, K, B, N, M,
                         hip.ptr(ws), ws.numel(), _st(x)),
                      "catre_op_linear_maxpool_bf16" if amp == 1 else "catre_op_linear_maxpool_split")
        elif fused:  # the max / arg-max is the GEMM's epilogue: the [rows, J] matrix never exists
            wp = torch.empty(J * K, dtype=torch.float32, device=x.device)
            hip.check(lib.catre_op_pack(hip.ptr(w2), w2.stride(0), J, K, 0, hip.ptr(wp), _st(x)), "catre_op_pack")
            need = lib.catre_op_linear_maxpool_ws_bytes(xc.shape[0], J)
            ws = _ws(need, x.device)
            hip.check(lib.catre_op_linear_maxpool(hip.ptr(xc), xc.stride(0), hip.ptr(wp), hip.ptr(b), hip.ptr(g),
                                                  hip.ptr(idx), J, K, B, N, M, hip.ptr(ws), ws.numel(), _st(x)),
                      "catre_op_linear_maxpool")
        else:
            y = _gemm_nt(xc, w2, b, False)
            hip.check(lib.catre_op_maxpool_fwd(hip.ptr(y), J, hip.ptr(g), hip.ptr(idx), J, B, N, M, _st(x)),
                      "catre_op_maxpool_fwd")
            del y
        if relu:
            g = _Relu.forward_only(g)
        ctx.save_for_backward(x, w, idx, g if relu else None)
        ctx.relu, ctx.dims = relu, (B, N, M, C, J, K)
        return g

    @staticmethod
    def backward(ctx, dg):
        x, w, idx, g = ctx.saved_tensors
        B, N, M, C, J, K = ctx.dims
        lib = hip.load()
        dg = _c(dg)
        if ctx.relu:
            t = torch.empty_like(dg)
            hip.check(lib.catre_op_relu_bwd(hip.ptr(dg), hip.ptr(g), hip.ptr(t), dg.numel(), _st(dg)), "catre_op_relu_bwd")
            dg = t
        w2 = _c(w.reshape(J, K))
        xc = _c(x)
        dw = torch.empty(J, K, dtype=torch.float32, device=x.device)
        db = torch.empty(J, dtype=torch.float32, device=x.device)
        hip.check(lib.catre_op_maxlin_bwd_w(hip.ptr(dg), hip.ptr(idx), hip.ptr(xc), xc.stride(0), hip.ptr(dw), hip.ptr(db),
                                            C, J, K, _st(x)), "catre_op_maxlin_bwd_w")
        dx = None
        if ctx.needs_input_grad[0]:
            if max(N, M) <= 4096 and J <= 1024 and K % 4 == 0 and K <= 512 and xc.shape[1] == K:
                dx = torch.empty_like(xc)  # every row is written (zeros where no channel had its maximum)
                hip.check(lib.catre_op_maxlin_bwd_x_rows(hip.ptr(dg), hip.ptr(idx), hip.ptr(w2), K, hip.ptr(dx),
                                                         dx.stride(0), J, K, B, N, M, _st(x)), "catre_op_maxlin_bwd_x_rows")
            else:
                dx = torch.zeros_like(xc)
                hip.check(lib.catre_op_maxlin_bwd_x(hip.ptr(dg), hip.ptr(idx), hip.ptr(w2), K, hip.ptr(dx), dx.stride(0), C,
                                                    J, K, _st(x)), "catre_op_maxlin_bwd_x")
        return dx, dw.reshape(w.shape), db, None, None, None, None, None


# ------------------------------------------------------------------------------------------------- row-sparse pooled chains
def _rows_compact(dg, idx, B, N, M):
    """Ascending list of the rows that are the arg-max of a channel with dg != 0, the dense -> compact map and the count -
    all on the device (no host sync): -> rows [R] int32, rowpos [R] int32, count [1] int32."""
    lib = hip.load()
    R, C = B * (N + M), dg.shape[0]
    dev = dg.device
    rows = torch.empty(R, dtype=torch.int32, device=dev)
    rowpos = torch.empty(R, dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(2 * C, dtype=torch.int32, device=dev)
    hip.check(lib.catre_op_rows_compact(hip.ptr(dg), hip.ptr(idx), dg.shape[1], B, N, M, hip.ptr(rows), hip.ptr(rowpos),
                                        hip.ptr(count), hip.ptr(scratch), _st(dg)), "catre_op_rows_compact")
    return rows, rowpos, count


def _gather_rows(src, rows, count):
    lib = hip.load()
    cap, cols = rows.shape[0], src.shape[1]
    dst = torch.empty(cap, cols, dtype=torch.float32, device=src.device)
    hip.check(lib.catre_op_gather_rows(hip.ptr(src), src.stride(0), hip.ptr(rows), hip.ptr(count), hip.ptr(dst), cols, cols,
                                       cap, _st(src)), "catre_op_gather_rows")
    return dst


def _scatter_rows(srcc, rowpos, cols):
    lib = hip.load()
    R = rowpos.shape[0]
    dst = torch.empty(R, cols, dtype=torch.float32, device=srcc.device)
    hip.check(lib.catre_op_scatter_rows(hip.ptr(srcc), srcc.stride(0), hip.ptr(rowpos), hip.ptr(dst), cols, cols, R,
                                        _st(srcc)), "catre_op_scatter_rows")
    return dst


def _dgrad_n(dy, w2, xmask, count, amp=0, mask=None, mask_rows=None):
    """dx[:n] = (dy[:n] .* (xmask[:n] > 0)) W  for the n = count[0] compact rows; w2: the layer's [J,K] weight; amp: the
    matrix pipe (0 fp32, 1 bf16 operands, 2 split) - the reduced-precision kernels want J in {64, 128, 256, 512}.
    mask / mask_rows (fp32 pipe only): dx row r is zeroed where row mask_rows[r] of the DENSE tensor `mask` is <= 0 - the
    ReLU of the layer in front, applied as the gradient is produced instead of on a gathered copy of its output."""
    lib = hip.load()
    cap, J = dy.shape
    K = w2.shape[1]
    assert _tiled_gemm_ok(cap, K, J, min_rows=0), (cap, K, J)
    dev = dy.device
    if amp in (1, 2) and J in (64, 128, 256, 512):
        wp = (_pack_bf16 if amp == 1 else _pack_split)(w2, K, J, dev, 1)
    else:
        amp = 0
        wp = torch.empty(J * K, dtype=torch.float32, device=dev)
        hip.check(lib.catre_op_pack(hip.ptr(w2), w2.stride(0), K, J, 1, hip.ptr(wp), _st(dy)), "catre_op_pack")
    dx = torch.empty(cap, K, dtype=torch.float32, device=dev)
    if mask is not None:
        assert mask.shape[1] == K
        # (bf16 rows: what the autocast encoder forward saves - only the bf16-operand kernel reads them)
        rows_bf16 = hip.ROWS_BF16 if mask.dtype == torch.bfloat16 else 0
        assert not rows_bf16 or amp == 1, "bf16 activation rows are read by the bf16-operand row GEMMs only"
        hip.check(lib.catre_op_gemm_rows_nr(hip.ptr(dy), dy.stride(0), hip.ptr(xmask),
                                            xmask.stride(0) if xmask is not None else 0, hip.ptr(wp), None, hip.ptr(mask),
                                            mask.stride(0), hip.ptr(mask_rows), hip.ptr(dx), K, cap, K, J, 0, hip.ptr(count),
                                            int(amp) | rows_bf16, _st(dy)), "catre_op_gemm_rows_nr")
        return dx
    hip.check(lib.catre_op_gemm_rows_n(hip.ptr(dy), dy.stride(0), hip.ptr(xmask), xmask.stride(0) if xmask is not None else 0,
                                       hip.ptr(wp), None, None, 0, hip.ptr(dx), K, cap, K, J, 0, hip.ptr(count), int(amp),
                                       _st(dy)), "catre_op_gemm_rows_n")
    return dx


def _wgrad_n(dy, x, ymask, count, amp=0, x_rows=None):
    """(dW [J,K], db [J]) = ((dy .* (ymask > 0))[:n]^T x[:n], column sums) over the n = count[0] compact rows.
    x_rows (fp32 pipe, or K <= 8): x is the DENSE tensor and compact row r pairs with its row x_rows[r]."""
    lib = hip.load()
    cap, J = dy.shape
    K = x.shape[1]
    assert J % 4 == 0 and K % 4 == 0
    buf = torch.empty(J * K + J, dtype=torch.float32, device=dy.device)
    dw, db = buf[: J * K].view(J, K), buf[J * K:]
    ws = _ws(lib.catre_op_gemm_tn_bias_ws_bytes(J, K, cap), dy.device)
    if x_rows is not None:
        rows_bf16 = hip.ROWS_BF16 if x.dtype == torch.bfloat16 else 0
        assert not rows_bf16 or amp == 1, "bf16 activation rows are read by the bf16-operand weight-gradient kernel only"
        hip.check(lib.catre_op_gemm_tn_bias_nr(hip.ptr(dy), dy.stride(0), hip.ptr(ymask),
                                               ymask.stride(0) if ymask is not None else 0, hip.ptr(x), x.stride(0),
                                               hip.ptr(x_rows), hip.ptr(dw), hip.ptr(db), J, K, cap, 0, hip.ptr(ws), ws.numel(),
                                               hip.ptr(count), int(amp) | rows_bf16, _st(dy)), "catre_op_gemm_tn_bias_nr")
        return dw, db
    assert x.dtype == torch.float32
    hip.check(lib.catre_op_gemm_tn_bias_n(hip.ptr(dy), dy.stride(0), hip.ptr(ymask), ymask.stride(0) if ymask is not None else 0,
                                          hip.ptr(x), x.stride(0), hip.ptr(dw), hip.ptr(db), J, K, cap, 0, hip.ptr(ws),
                                          ws.numel(), hip.ptr(count), int(amp), _st(dy)), "catre_op_gemm_tn_bias_n")
    return dw, db


class _PooledChain(torch.autograd.Function):
    """x -> y1 = relu(x W1^T + b1) -> y2 = relu(y1 W2^T + b2) -> g = max over the points of each cloud of (y2 W3^T + b3)
    (-> ReLU): one conv stack of the encoder in front of its max-pool (pointnet.py:24-28, 57-61, 112-116) as ONE graph node
    around outputs the fused forward kernels have already written (y1, y2, g, idx).

    The backward is ROW-SPARSE.  Only the arg-max row of a (cloud, channel) carries gradient into y2, so dy2 - and with it
    dy1 and dx - is zero on every row that is nobody's arg-max: ~70 % of them at N = M = 1024.  The live rows are compacted
    on the device (ascending; the count never reaches the host) and the two dgrad and two wgrad GEMMs run on those rows
    only; dx is scattered back to dense rows (zeros elsewhere), the weight gradients are sums over the same rows in the
    same order as before minus exact zeros.  The GEMMs run on the matrix pipe of the mode the forward ran in (fp32, bf16
    operands under autocast, split); the pooled layer's own sparse gathers are fp32 in every mode, as before."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, relu_pool, B, N, M, y1, y2, g, idx):
        gout = _Relu.forward_only(g) if relu_pool else g
        # y1 is y2 is None: the forward kernel stored no activation rows - the backward recomputes them on its live rows
        ctx.save_for_backward(x, w1, w2, w3, y1, y2, idx, gout if relu_pool else None, b1 if y1 is None else None,
                              b2 if y1 is None else None)
        ctx.dims, ctx.relu_pool, ctx.amp = (B, N, M), relu_pool, _amp()
        ctx.has_b = (b1 is not None, b2 is not None, b3 is not None)
        return gout

    @staticmethod
    def backward(ctx, dg):
        return _pooled_chain_backward(ctx, dg, None) + (None,) * 8


def _pooled_chain_backward(ctx, dg, merge):
    """Shared by _PooledChain and _PointfeatHub.  merge = (dobj, dmax, idx_max) or None: the other gradients x receives -
    from the rotation heads (object-major rows) and from max over points - added while dx is scattered to dense rows.
    -> (dx, dw1, db1, dw2, db2, dw3, db3)."""
    x, w1, w2, w3, y1, y2, idx, gout = ctx.saved_tensors[:8]
    if y1 is None:
        return _pooled_chain_backward_recompute(ctx, dg)
    B, N, M = ctx.dims
    lib = hip.load()
    dg = _c(dg)
    if ctx.relu_pool:
        t = torch.empty_like(dg)
        hip.check(lib.catre_op_relu_bwd(hip.ptr(dg), hip.ptr(gout), hip.ptr(t), dg.numel(), _st(dg)), "catre_op_relu_bwd")
        dg = t
    C, J3 = dg.shape
    w1m, w2m, w3m = (_c(w.reshape(w.shape[0], -1)) for w in (w1, w2, w3))
    K3 = w3m.shape[1]
    y1, y2 = _c(y1), _c(y2)
    # pooled layer: its own weight / bias gradient gathers the arg-max rows of y2 (unchanged)
    dw3 = torch.empty(J3, K3, dtype=torch.float32, device=dg.device)
    db3 = torch.empty(J3, dtype=torch.float32, device=dg.device)
    # (y1 / y2 as bf16 rows: what the autocast encoder forward saves - the `_h` forms of the two pooled-layer kernels and
    # the CATRE_ROWS_BF16 flag of the row GEMMs below read them in place)
    hb = y2.dtype == torch.bfloat16
    assert hb == (y1.dtype == torch.bfloat16)
    fn_w, fn_x = ((lib.catre_op_maxlin_bwd_w_h, lib.catre_op_maxlin_bwd_x_compact_h) if hb
                  else (lib.catre_op_maxlin_bwd_w, lib.catre_op_maxlin_bwd_x_compact))
    hip.check(fn_w(hip.ptr(dg), hip.ptr(idx), hip.ptr(y2), y2.stride(0), hip.ptr(dw3), hip.ptr(db3), C, J3, K3, _st(dg)),
              "catre_op_maxlin_bwd_w")
    rows, rowpos, count = _rows_compact(dg, idx, B, N, M)
    cap = rows.shape[0]
    # dy2 on the live rows, with y2's own ReLU applied on the way out
    dy2 = torch.empty(cap, K3, dtype=torch.float32, device=dg.device)
    hip.check(fn_x(hip.ptr(dg), hip.ptr(idx), hip.ptr(w3m), K3, hip.ptr(rowpos), hip.ptr(y2), y2.stride(0), hip.ptr(dy2), K3,
                   J3, K3, B, N, M, _st(dg)), "catre_op_maxlin_bwd_x_compact")
    amp = ctx.amp
    xk = _c(_pad_cols(x, 4))
    # no gathered copies of the saved activations: the weight-gradient GEMMs read the dense y1 / x through the live-row list,
    # and y1's ReLU is applied where dy1 is produced (mask rows through the same list) - every compute mode
    dw2, db2 = _wgrad_n(dy2, y1, None, count, amp, x_rows=rows)
    dy1 = _dgrad_n(dy2, w2m, None, count, amp, mask=y1, mask_rows=rows)     # [cap, K2], y1's ReLU applied
    dw1, db1 = _wgrad_n(dy1, xk, None, count, amp, x_rows=rows)
    dw1 = dw1[:, : w1m.shape[1]]
    dx = None
    if ctx.needs_input_grad[0]:
        dxc = _dgrad_n(dy1, w1m, None, count, amp)                  # [cap, K1]
        if merge is None:
            dx = _scatter_rows(dxc, rowpos, dxc.shape[1])
        else:
            dobj, dmax, idx_max = merge
            cols = dxc.shape[1]
            dx = torch.empty(rowpos.shape[0], cols, dtype=torch.float32, device=dg.device)
            dobj = _c(dobj) if dobj is not None else None
            dmax = _c(dmax) if dmax is not None else None
            hip.check(lib.catre_op_scatter_rows_merge(
                hip.ptr(dxc), dxc.stride(0), hip.ptr(rowpos), hip.ptr(dobj), dobj.stride(0) if dobj is not None else 0,
                hip.ptr(dmax), hip.ptr(idx_max), dmax.shape[1] if dmax is not None else 0, hip.ptr(dx), cols, cols, B, N, M,
                _st(dg)), "catre_op_scatter_rows_merge")
        if dx.shape[1] != x.shape[1]:
            dx = _c(dx[:, : x.shape[1]])
    hb = ctx.has_b
    return (dx, _c(dw1).reshape(w1.shape), _c(db1) if hb[0] else None, _c(dw2).reshape(w2.shape),
            _c(db2) if hb[1] else None, dw3.reshape(w3.shape), db3 if hb[2] else None)


def _pooled_chain_backward_recompute(ctx, dg):
    """:func:`_pooled_chain_backward` for a chain whose forward stored no activation rows (the STN stacks, fp32): the live
    rows' y1 / y2 are rebuilt as COMPACT rows by the forward kernels' own device code (`catre_op_stn_recompute`: same bits),
    and every consumer below reads compact operands - no row indirection except into the chain's input."""
    x, w1, w2, w3, _, _, idx, gout, b1, b2 = ctx.saved_tensors[:10]
    B, N, M = ctx.dims
    lib = hip.load()
    dev = dg.device
    dg = _c(dg)
    if ctx.relu_pool:
        t = torch.empty_like(dg)
        hip.check(lib.catre_op_relu_bwd(hip.ptr(dg), hip.ptr(gout), hip.ptr(t), dg.numel(), _st(dg)), "catre_op_relu_bwd")
        dg = t
    C, J3 = dg.shape
    w1m, w2m, w3m = (_c(w.reshape(w.shape[0], -1)) for w in (w1, w2, w3))
    K3 = w3m.shape[1]
    rows, rowpos, count = _rows_compact(dg, idx, B, N, M)
    cap = rows.shape[0]
    xc = _c(x)
    kind = 0 if w1m.shape[1] <= 4 else 1
    wp2 = torch.empty(w2m.numel(), dtype=torch.float32, device=dev)
    hip.check(lib.catre_op_pack(hip.ptr(w2m), w2m.stride(0), w2m.shape[0], w2m.shape[1], 0, hip.ptr(wp2), _st(dg)), "catre_op_pack")
    wp1 = None
    if kind == 1:
        wp1 = torch.empty(w1m.numel(), dtype=torch.float32, device=dev)
        hip.check(lib.catre_op_pack(hip.ptr(w1m), w1m.stride(0), w1m.shape[0], w1m.shape[1], 0, hip.ptr(wp1), _st(dg)),
                  "catre_op_pack")
    y1c = torch.empty(cap, w1m.shape[0], dtype=torch.float32, device=dev)
    y2c = torch.empty(cap, w2m.shape[0], dtype=torch.float32, device=dev)
    hip.check(lib.catre_op_stn_recompute(kind, hip.ptr(xc), xc.stride(0), hip.ptr(rows), hip.ptr(count),
                                         hip.ptr(w1m) if kind == 0 else None, hip.ptr(wp1), hip.ptr(b1), hip.ptr(wp2),
                                         hip.ptr(b2), hip.ptr(y1c), hip.ptr(y2c), cap, _st(dg)), "catre_op_stn_recompute")
    # pooled layer: weight / bias gradient from the arg-max rows of y2 (compact)
    dw3 = torch.empty(J3, K3, dtype=torch.float32, device=dev)
    db3 = torch.empty(J3, dtype=torch.float32, device=dev)
    hip.check(lib.catre_op_maxlin_bwd_w_c(hip.ptr(dg), hip.ptr(idx), hip.ptr(rowpos), hip.ptr(y2c), y2c.stride(0), hip.ptr(dw3),
                                          hip.ptr(db3), C, J3, K3, _st(dg)), "catre_op_maxlin_bwd_w_c")
    dy2 = torch.empty(cap, K3, dtype=torch.float32, device=dev)
    hip.check(lib.catre_op_maxlin_bwd_x_compact_cm(hip.ptr(dg), hip.ptr(idx), hip.ptr(w3m), K3, hip.ptr(rowpos), hip.ptr(y2c),
                                                   y2c.stride(0), hip.ptr(dy2), K3, J3, K3, B, N, M, _st(dg)),
              "catre_op_maxlin_bwd_x_compact_cm")
    amp = ctx.amp
    xk = _c(_pad_cols(x, 4))
    dw2, db2 = _wgrad_n(dy2, y1c, None, count, amp)
    dy1 = _dgrad_n_masked(dy2, w2m, count, y1c)                              # [cap, K2], y1's ReLU applied
    dw1, db1 = _wgrad_n(dy1, xk, None, count, amp, x_rows=rows)
    dw1 = dw1[:, : w1m.shape[1]]
    dx = None
    if ctx.needs_input_grad[0]:
        dxc = _dgrad_n(dy1, w1m, None, count, amp)
        dx = _scatter_rows(dxc, rowpos, dxc.shape[1])
        if dx.shape[1] != x.shape[1]:
            dx = _c(dx[:, : x.shape[1]])
    hb = ctx.has_b
    return (dx, _c(dw1).reshape(w1.shape), _c(db1) if hb[0] else None, _c(dw2).reshape(w2.shape),
            _c(db2) if hb[1] else None, dw3.reshape(w3.shape), db3 if hb[2] else None)


def _dgrad_n_masked(dy, w2, count, mask_c):
    """dx[:n] = dy[:n] W, zeroed where the COMPACT mask rows are <= 0 (fp32 pipe)."""
    lib = hip.load()
    cap, J = dy.shape
    K = w2.shape[1]
    wp = torch.empty(J * K, dtype=torch.float32, device=dy.device)
    hip.check(lib.catre_op_pack(hip.ptr(w2), w2.stride(0), K, J, 1, hip.ptr(wp), _st(dy)), "catre_op_pack")
    dx = torch.empty(cap, K, dtype=torch.float32, device=dy.device)
    hip.check(lib.catre_op_gemm_rows_n(hip.ptr(dy), dy.stride(0), None, 0, hip.ptr(wp), None, hip.ptr(mask_c), mask_c.stride(0),
                                       hip.ptr(dx), K, cap, K, J, 0, hip.ptr(count), 0, _st(dy)), "catre_op_gemm_rows_n")
    return dx


class _PointfeatHub(torch.autograd.Function):
    """The trunk's conv stack (_PooledChain on x = pointfeat) TOGETHER with pointfeat's two other consumers - max over the
    points of each cloud (the tail of flat_pcl_feat, CATRE_disR_shared.py:69) and the object-major copy the rotation heads
    read (:86) - as one node: -> (g, pfmax, pf_obj).  Its backward adds the three gradients pointfeat receives while the
    row-sparse one is scattered back to dense rows: one pass over [rows, 64] instead of a zero-fill + scatter, two strided
    copies + cat, and two full-size adds by autograd."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, B, N, M, y1, y2, g, idx, obj_copy=True):
        lib = hip.load()
        xc = _c(x)
        J, C = xc.shape[1], (2 * B if M > 0 else B)
        pfmax = torch.empty(C, J, dtype=torch.float32, device=x.device)
        idx_max = torch.empty(C, J, dtype=torch.int32, device=x.device)
        hip.check(lib.catre_op_maxpool_fwd(hip.ptr(xc), J, hip.ptr(pfmax), hip.ptr(idx_max), J, B, N, M, _st(x)),
                  "catre_op_maxpool_fwd")
        if obj_copy:
            pf_obj = torch.cat([xc[: B * N].view(B, N, J), xc[B * N:].view(B, M, J)], 1).reshape(B * (N + M), J)
        else:
            # the consumer (train_ops._RotHeads) reads pointfeat in its cloud-major order: no copy, the third output is the
            # input's buffer; its GRADIENT still arrives object-major, like the copy's would
            pf_obj = xc.detach()
        ctx.save_for_backward(x, w1, w2, w3, y1, y2, idx, None, idx_max)
        ctx.dims, ctx.relu_pool, ctx.amp = (B, N, M), False, _amp()
        ctx.has_b = (b1 is not None, b2 is not None, b3 is not None)
        return g, pfmax, pf_obj

    @staticmethod
    def backward(ctx, dg, dmax, dobj):
        idx_max = ctx.saved_tensors[8]
        if dg is None:
            raise RuntimeError("pointfeat hub: the pooled trunk feature received no gradient")
        return _pooled_chain_backward(ctx, dg, (dobj, dmax, idx_max)) + (None,) * 8


def pointfeat_hub(x, w1, b1, w2, b2, w3, b3, B, N, M, pre, obj_copy=True):
    """obj_copy=False: the third output is pointfeat itself (cloud-major rows, no copy) - for a consumer that reads it in
    that order and returns an object-major gradient (the fused fp32 rotation heads)."""
    y1, y2, g, idx = pre
    return _PointfeatHub.apply(x, w1, b1, w2, b2, w3, b3, B, N, M, y1, y2, g, idx, obj_copy)


def pooled_chain_ok(x, w1, w2, w3, N, M):
    """Shapes the row-sparse chain takes (the encoder's three conv stacks at N, M multiples of 64; every compute mode)."""
    if N % 64 or M % 64 or max(N, M) > 4096:
        return False
    j1, j2, j3 = w1.shape[0], w2.shape[0], w3.shape[0]
    k1 = w1.reshape(j1, -1).shape[1]
    if x.requires_grad and not _tiled_gemm_ok(1, k1, j1, min_rows=0):
        return False  # the chain's input gradient is a tiled row GEMM with K1 outputs: a differentiable 3-d input takes the layer-wise ops
    return (j1 in (64, 128) and j2 in (128, 512) and j3 <= 1024 and j3 % 32 == 0 and (k1 <= 8 or k1 in (64, 128)))


def pooled_chain(x, w1, b1, w2, b2, w3, b3, relu_pool, B, N, M, pre):
    """pre = (y1, y2, g, idx) written by the fused forward kernel."""
    y1, y2, g, idx = pre
    return _PooledChain.apply(x, w1, b1, w2, b2, w3, b3, relu_pool, B, N, M, y1, y2, g, idx)


class _Relu:
    @staticmethod
    def forward_only(g):
        # relu on the tiny pooled tensor: relu_bwd(dy=g, y=g) == where(g > 0, g, 0)
        lib = hip.load()
        out = torch.empty_like(g)
        hip.check(lib.catre_op_relu_bwd(hip.ptr(g), hip.ptr(g), hip.ptr(out), g.numel(), _st(g)), "catre_op_relu_bwd")
        return out


def linear_maxpool(x, w, b, relu, B, N, M, pre=None):
    return _LinearMaxPool.apply(x, w, b, relu, B, N, M, pre)


class _MaxPool(torch.autograd.Function):
    """Plain max over the points of each cloud (dense scatter backward) - used for max_n pointfeat."""

    @staticmethod
    def forward(ctx, y, B, N, M):
        lib = hip.load()
        y = _c(y)
        J = y.shape[1]
        C = 2 * B if M > 0 else B
        g = torch.empty(C, J, dtype=torch.float32, device=y.device)
        idx = torch.empty(C, J, dtype=torch.int32, device=y.device)
        hip.check(lib.catre_op_maxpool_fwd(hip.ptr(y), J, hip.ptr(g), hip.ptr(idx), J, B, N, M, _st(y)), "catre_op_maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = tuple(y.shape)
        return g

    @staticmethod
    def backward(ctx, dg):
        (idx,) = ctx.saved_tensors
        lib = hip.load()
        dg = _c(dg)
        dy = torch.zeros(ctx.shape, dtype=torch.float32, device=dg.device)
        hip.check(lib.catre_op_maxpool_scatter(hip.ptr(dg), hip.ptr(idx), hip.ptr(dy), ctx.shape[1], dg.shape[0],
                                               dg.shape[1], _st(dg)), "catre_op_maxpool_scatter")
        return dy, None, None, None


def maxpool_points(y, B, N, M):
    return _MaxPool.apply(y, B, N, M)


# ------------------------------------------------------------------------------------------------- per-cloud transforms
class _CloudMatmul(torch.autograd.Function):
    """y[r,:] = x[r,:kd] T[cloud(r)]; output has `out_cols` columns (zero beyond kd) so it can feed the MFMA GEMM."""

    @staticmethod
    def forward(ctx, x, T, B, N, M, out_cols, pre):
        lib = hip.load()
        kd = T.shape[-1]
        x, T = _c(x), _c(T)
        R = x.shape[0]
        if pre is not None:
            y = pre
        else:
            y = torch.zeros(R, out_cols, dtype=torch.float32, device=x.device) if out_cols != kd else \
                torch.empty(R, kd, dtype=torch.float32, device=x.device)
            hip.check(lib.catre_op_cloud_matmul(hip.ptr(x), x.stride(0), hip.ptr(T), hip.ptr(y), out_cols, kd, B, N, M, 0,
                                                _st(x)), "catre_op_cloud_matmul")
        ctx.save_for_backward(x, T)
        ctx.dims = (B, N, M, kd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, T = ctx.saved_tensors
        B, N, M, kd = ctx.dims
        lib = hip.load()
        dy = _c(dy)
        dx = dT = None
        if ctx.needs_input_grad[0]:
            dx = torch.zeros_like(x) if x.shape[1] != kd else torch.empty_like(x)
            hip.check(lib.catre_op_cloud_matmul(hip.ptr(dy), dy.stride(0), hip.ptr(T), hip.ptr(dx), dx.stride(0), kd, B, N,
                                                M, 1, _st(x)), "catre_op_cloud_matmul^T")
        if ctx.needs_input_grad[1]:
            dT = torch.empty_like(T)
            hip.check(lib.catre_op_cloud_matmul_bwd_t(hip.ptr(x), x.stride(0), hip.ptr(dy), dy.stride(0), hip.ptr(dT), kd,
                                                      B, N, M, _st(x)), "catre_op_cloud_matmul_bwd_t")
        return dx, dT, None, None, None, None, None


def cloud_matmul(x, T, B, N, M, out_cols=None, pre=None):
    return _CloudMatmul.apply(x, T, B, N, M, T.shape[-1] if out_cols is None else out_cols, pre)


# ------------------------------------------------------------------------------------------------- row re-ordering
class _ObjectMajor(torch.autograd.Function):
    """cloud-major rows (B*N observed, then B*M prior) -> object-major rows ([N observed | M prior] per object): the
    order the rotation heads want (``cat(pcl_feat, kps_feat, dim=2)``, CATRE_disR_shared.py:86).  One gather copy each
    way - autograd's own slice / cat backward would zero-fill and accumulate two full-size tensors."""

    @staticmethod
    def forward(ctx, x, B, N, M):
        C = x.shape[1]
        ctx.dims = (B, N, M, C)
        return torch.cat([x[: B * N].view(B, N, C), x[B * N:].view(B, M, C)], 1).reshape(B * (N + M), C)

    @staticmethod
    def backward(ctx, d):
        B, N, M, C = ctx.dims
        d = _c(d).view(B, N + M, C)
        return torch.cat([d[:, :N].reshape(B * N, C), d[:, N:].reshape(B * M, C)], 0), None, None, None


def object_major(x, B, N, M):
    return _ObjectMajor.apply(x, B, N, M)


# ------------------------------------------------------------------------------------------------- per-cloud bias
class _RowBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, bias, B, N, M):
        lib = hip.load()
        # in place when y is an intermediate nobody else holds (the layer-0 GEMM output: a plain linear does not save
        # its result) - a [B*(N+M), 256] clone is 1 GiB of traffic per head
        if y.requires_grad and not y.is_leaf and y.is_contiguous():
            ctx.mark_dirty(y)
        else:
            y = y.clone()
        bias = _c(bias)
        hip.check(lib.catre_op_rowbias_add(hip.ptr(y), y.stride(0), hip.ptr(bias), y.shape[1], B, N, M, _st(y)),
                  "catre_op_rowbias_add")
        ctx.dims = (B, N, M)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, N, M = ctx.dims
        lib = hip.load()
        dy = _c(dy)
        J = dy.shape[1]
        db = torch.empty(2 * B if M > 0 else B, J, dtype=torch.float32, device=dy.device)
        hip.check(lib.catre_op_rowbias_bwd(hip.ptr(dy), dy.stride(0), hip.ptr(db), J, B, N, M, _st(dy)), "catre_op_rowbias_bwd")
        return dy, db, None, None, None


class _RotLinear(torch.autograd.Function):
    """The rot-head linears in one kernel each: y = x W^T + bias, where the bias is per channel ([J]) or per cloud
    ([2B,J]: layer 0, whose global-feature half is a bias per cloud), and - J == 256 - the per-64-row-tile GroupNorm
    partials of y come out of the same epilogue (second, non-differentiable output; gn_points_gelu takes them instead of
    a statistics pass over y).  Backward: dbias = column sums of dy (per cloud or total), dx / dW like a plain linear."""

    @staticmethod
    def forward(ctx, x, w, bias, per_cloud, B, N, M, pre=None):
        lib = hip.load()
        w2 = _c(w.reshape(w.shape[0], -1))
        J, K = w2.shape
        amp = _amp()
        if amp and K not in (64, 128, 256, 512):
            amp = 0
        if pre is not None:  # (y, partials) already computed by a fused forward (rot_heads_forward): only the graph node
            y, part = pre
            ctx.save_for_backward(x, w)
            ctx.dims, ctx.amp, ctx.per_cloud, ctx.has_b = (B, N, M), amp, bool(per_cloud), bias is not None
            part = part if part is not None else torch.empty(0, device=x.device)
            ctx.mark_non_differentiable(part)
            return y, part
        xc, bc = _c(x), (_c(bias) if bias is not None else None)
        if amp == 1:
            wp = _pack_bf16(w2, J, K, x.device)
        elif amp == 2:
            wp = _pack_split(w2, J, K, x.device)
        else:
            wp = torch.empty(J * K, dtype=torch.float32, device=x.device)
            hip.check(lib.catre_op_pack(hip.ptr(w2), w2.stride(0), J, K, 0, hip.ptr(wp), _st(x)), "catre_op_pack")
        R = xc.shape[0]
        y = torch.empty(R, J, dtype=torch.float32, device=x.device)
        part = torch.empty(R // 64, 32, 2, dtype=torch.float32, device=x.device) if J == 256 else None
        hip.check(lib.catre_op_gemm_rows_gn(hip.ptr(xc), xc.stride(0), hip.ptr(wp), hip.ptr(bc), int(per_cloud), hip.ptr(y), J,
                                            J, K, B, N, M, hip.ptr(part), amp, _st(x)), "catre_op_gemm_rows_gn")
        ctx.save_for_backward(x, w)
        ctx.dims, ctx.amp, ctx.per_cloud, ctx.has_b = (B, N, M), amp, bool(per_cloud), bias is not None
        if part is None:
            part = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(part)
        return y, part

    @staticmethod
    def backward(ctx, dy, _dpart):
        x, w = ctx.saved_tensors
        B, N, M = ctx.dims
        lib = hip.load()
        dy = _c(dy)
        J = dy.shape[1]
        w2 = w.reshape(w.shape[0], -1)
        dx = dw = db = None
        want_db = ctx.has_b and ctx.needs_input_grad[2]
        if want_db and ctx.per_cloud:
            db = torch.empty(2 * B if M > 0 else B, J, dtype=torch.float32, device=dy.device)
            hip.check(lib.catre_op_rowbias_bwd(hip.ptr(dy), dy.stride(0), hip.ptr(db), J, B, N, M, _st(dy)),
                      "catre_op_rowbias_bwd")
        if ctx.needs_input_grad[0]:
            dx = _gemm_nt(dy, None, None, False, amp=ctx.amp, wT=_c(w2))
        if ctx.needs_input_grad[1]:
            if want_db and not ctx.per_cloud:
                dw, db = _gemm_tn(dy, _c(x), with_bias=True, amp=ctx.amp)
                db = _c(db)
            else:
                dw = _gemm_tn(dy, _c(x), amp=ctx.amp)
            dw = _c(dw).reshape(w.shape)
        elif want_db and not ctx.per_cloud:
            db = _colsum(dy)
        return dx, dw, db, None, None, None, None, None


def _rot_linear_ok(R, J, K, N, M):
    return N % 64 == 0 and M % 64 == 0 and _tiled_gemm_ok(R, J, K, min_rows=64) and K % 8 == 0


def linear_cloudbias(x, w, bias, B, N, M, with_gn_partials=False, pre=None):
    """x [B*(N+M), K] object-major rows, w [J,K], bias [2B,J] -> x w^T + bias[cloud(row)].  One fused kernel when the
    tiles cannot straddle clouds (N, M multiples of 64) and the shape is tiled; else linear + rowbias_add.  With
    with_gn_partials returns (y, partials or None)."""
    J, K = w.shape[0], w.reshape(w.shape[0], -1).shape[1]
    if _rot_linear_ok(x.shape[0], J, K, N, M):
        y, part = _RotLinear.apply(x, w, bias, True, B, N, M, pre)
        return (y, part if part.numel() else None) if with_gn_partials else y
    assert pre is None
    y = rowbias_add(linear(x, w, None), bias, B, N, M)
    return (y, None) if with_gn_partials else y


def linear_gn_partials(x, w, bias, B, N, M, pre=None):
    """y = x w^T + bias plus the GroupNorm(32,256) tile partials of y from the same kernel -> (y, partials or None)."""
    J, K = w.shape[0], w.reshape(w.shape[0], -1).shape[1]
    if J == 256 and _rot_linear_ok(x.shape[0], J, K, N, M):
        y, part = _RotLinear.apply(x, w, bias, False, B, N, M, pre)
        return y, part
    assert pre is None
    return linear(x, w, bias), None


def rowbias_add(y, bias, B, N, M):
    """y [B*(N+M), J] object-major += bias[cloud] (bias [2B, J])."""
    return _RowBias.apply(y, bias, B, N, M)


# ------------------------------------------------------------------------------------------------- GroupNorm + GELU
class _GNPointsGelu(torch.autograd.Function):
    """gelu(GroupNorm(32,256)(y)) with statistics over the P points of each object (rows object-major)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, B, P, part, pre=None):
        lib = hip.load()
        y = _c(y)
        if pre is not None:  # (a, stat) from a fused forward
            a, stat = pre
            ctx.save_for_backward(y, gamma, beta, stat)
            ctx.dims = (B, P)
            return a
        a = torch.empty_like(y)
        stat = torch.empty(B, 32, 2, dtype=torch.float32, device=y.device)
        if part is not None and P % 64 == 0:  # statistics from the producing GEMM's tile partials: no pass over y
            hip.check(lib.catre_op_gnp_gelu_fwd_pre(hip.ptr(y), hip.ptr(part), hip.ptr(gamma), hip.ptr(beta), hip.ptr(a),
                                                    hip.ptr(stat), B, P, _st(y)), "catre_op_gnp_gelu_fwd_pre")
        else:
            hip.check(lib.catre_op_gnp_gelu_fwd(hip.ptr(y), hip.ptr(gamma), hip.ptr(beta), hip.ptr(a), hip.ptr(stat), B, P,
                                                _st(y)), "catre_op_gnp_gelu_fwd")
        ctx.save_for_backward(y, gamma, beta, stat)
        ctx.dims = (B, P)
        return a

    @staticmethod
    def backward(ctx, da):
        y, gamma, beta, stat = ctx.saved_tensors
        B, P = ctx.dims
        lib = hip.load()
        da = _c(da)
        dy = torch.empty_like(y)
        dg, db = torch.empty_like(gamma), torch.empty_like(beta)
        nch = (P + 127) // 128
        ws = _ws((B * 64 + B * nch * (64 + 512) + (B * nch // 64 + 1) * 512) * 4, y.device)
        hip.check(lib.catre_op_gnp_gelu_bwd(hip.ptr(da), hip.ptr(y), hip.ptr(stat), hip.ptr(gamma), hip.ptr(beta),
                                            hip.ptr(dy), hip.ptr(dg), hip.ptr(db), 0, hip.ptr(ws), ws.numel(), B, P,
                                            _st(y)), "catre_op_gnp_gelu_bwd")
        return dy, dg, db, None, None, None, None


def gn_points_gelu(y, gamma, beta, B, P, part=None, pre=None):
    """part: optional per-64-row-tile GroupNorm partials of y ([B*P/64, 32, 2], from linear_cloudbias /
    linear_gn_partials).  pre: (a, stat [B,32,2]) already computed by a fused forward."""
    return _GNPointsGelu.apply(y, gamma, beta, B, P, part, pre)


class _GNPointsGeluNeck(torch.autograd.Function):
    """neck(gelu(GroupNorm(32,256)(y))) -> [B*P, 3] in one op (RotHead's last stage, conv_out_per_rot_head.py:132-137):
    the [B*P,256] activation between GroupNorm/GELU and the 256 -> rot_dim conv is neither stored forward nor
    differentiated through backward - d a = dy3 Wn is rebuilt from the three floats of its row."""

    @staticmethod
    def forward(ctx, y, gamma, beta, wn, bn, B, P, part):
        lib = hip.load()
        y, wn = _c(y), _c(wn)
        bnc = _c(bn) if bn is not None else None
        y3 = torch.empty(y.shape[0], 3, dtype=torch.float32, device=y.device)
        stat = torch.empty(B, 32, 2, dtype=torch.float32, device=y.device)
        hip.check(lib.catre_op_gnp_gelu_neck_fwd(hip.ptr(y), hip.ptr(part), hip.ptr(gamma), hip.ptr(beta), hip.ptr(wn),
                                                 hip.ptr(bnc), hip.ptr(y3), hip.ptr(stat), B, P, _st(y)),
                  "catre_op_gnp_gelu_neck_fwd")
        ctx.save_for_backward(y, gamma, beta, wn, stat)
        ctx.dims, ctx.has_bn = (B, P), bn is not None
        return y3

    @staticmethod
    def backward(ctx, dy3):
        y, gamma, beta, wn, stat = ctx.saved_tensors
        B, P = ctx.dims
        lib = hip.load()
        dy3 = _c(dy3)
        dy = torch.empty_like(y)
        dpar = torch.empty(5, 256, dtype=torch.float32, device=y.device)
        ws = _ws(lib.catre_op_gnp_gelu_neck_bwd_ws_bytes(B, P), y.device)
        hip.check(lib.catre_op_gnp_gelu_neck_bwd(hip.ptr(dy3), hip.ptr(y), hip.ptr(stat), hip.ptr(gamma), hip.ptr(beta),
                                                 hip.ptr(wn), hip.ptr(dy), hip.ptr(dpar), 0, hip.ptr(ws), ws.numel(), B, P,
                                                 _st(y)), "catre_op_gnp_gelu_neck_bwd")
        dbn = _colsum(dy3) if ctx.has_bn else None
        return dy, dpar[0], dpar[1], dpar[2:5], dbn, None, None, None


def gn_points_gelu_neck(y, gamma, beta, wn, bn, B, P, part):
    """y [B*P,256] with its tile partials ``part`` (linear_gn_partials), wn [3,256], bn [3] or None -> [B*P,3]."""
    return _GNPointsGeluNeck.apply(y, gamma, beta, wn, bn, B, P, part)


class _NeckTail(torch.autograd.Function):
    """RotHead's tail - GroupNorm(32,256), GELU, neck conv 256 -> rot_dim and conv_p (the weighted sum over the points),
    conv_out_per_rot_head.py:132-140 - as ONE node: y [B*P,256] (+ its tile partials) -> out [B,3].  Because the node ends
    behind conv_p, every row's neck gradient is wp[p] * dout[b]: the forward also leaves three per-channel moments per tile
    (catre_op_gnp_gelu_neck_fwd_s) and the backward takes the GroupNorm sums, dgamma, dbeta and dWn from them
    (catre_op_gnp_gelu_neck_bwd_s) instead of a reduction pass over y (0.5 GiB per head).  Every compute mode: the block is
    fp32 arithmetic in all of them; the fused fp32 heads (_RotHeads) have the same thing built in."""

    @staticmethod
    def forward(ctx, y, gamma, beta, wn, bn, wp, bp, B, P, part):
        lib = hip.load()
        y, wn = _c(y), _c(wn)
        bnc = _c(bn) if bn is not None else None
        wv = _c(wp.reshape(-1))
        dev = y.device
        y3 = torch.empty(y.shape[0], 3, dtype=torch.float32, device=dev)
        stat = torch.empty(B, 32, 2, dtype=torch.float32, device=dev)
        spart = torch.empty(y.shape[0] // 64, 3, 256, dtype=torch.float32, device=dev)
        hip.check(lib.catre_op_gnp_gelu_neck_fwd_s(hip.ptr(y), hip.ptr(part), hip.ptr(gamma), hip.ptr(beta), hip.ptr(wn),
                                                   hip.ptr(bnc), hip.ptr(wv), hip.ptr(y3), hip.ptr(stat), hip.ptr(spart), B, P,
                                                   _st(y)), "catre_op_gnp_gelu_neck_fwd_s")
        out = torch.empty(B, 3, dtype=torch.float32, device=dev)
        hip.check(lib.catre_op_wsum_fwd(hip.ptr(y3), hip.ptr(wv), hip.ptr(bp), hip.ptr(out), B, P, _st(y)), "catre_op_wsum_fwd")
        ctx.save_for_backward(y, gamma, beta, wn, stat, spart, y3, wv)
        ctx.dims, ctx.has_bn, ctx.has_bp, ctx.wp_shape = (B, P), bn is not None, bp is not None, wp.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, beta, wn, stat, spart, y3, wv = ctx.saved_tensors
        B, P = ctx.dims
        lib = hip.load()
        dev = y.device
        dout = _c(dout)
        dy3, dwp, dbp, dbn = _wsum_backward(dout, y3, wv, B, P, ctx.has_bp, ctx.has_bn)
        dy = torch.empty_like(y)
        dpar = torch.empty(5, 256, dtype=torch.float32, device=dev)
        ws = _ws(lib.catre_op_gnp_gelu_neck_bwd_ws_bytes(B, P), dev)
        hip.check(lib.catre_op_gnp_gelu_neck_bwd_s(hip.ptr(dy3), hip.ptr(dout), hip.ptr(spart), hip.ptr(y), hip.ptr(stat),
                                                   hip.ptr(gamma), hip.ptr(beta), hip.ptr(wn), hip.ptr(dy), hip.ptr(dpar),
                                                   hip.ptr(ws), ws.numel(), B, P, _st(dout)), "catre_op_gnp_gelu_neck_bwd_s")
        return dy, dpar[0], dpar[1], dpar[2:5], dbn, dwp.view(ctx.wp_shape), dbp, None, None, None


def neck_tail(y, gamma, beta, wn, bn, wp, bp, B, P, part):
    """y [B*P,256] with its tile partials ``part`` (linear_gn_partials; P % 64 == 0), wn [3,256], bn [3] or None, conv_p weight
    [1,P,1] and bias -> [B,3] (columns >= rot_dim are zero)."""
    return _NeckTail.apply(y, gamma, beta, wn, bn, wp, bp, B, P, part)


class _RotL1TailLP(torch.autograd.Function):
    """A RotHead from its second linear to the conv_p output under torch.autocast - Conv1d(256 -> 256) on bf16 operands,
    GroupNorm(32,256), GELU, neck, conv_p (conv_out_per_rot_head.py:129-140) - as ONE node: a [B*P,256] -> out [B,3].
    Forward: the bf16-operand row GEMM with the GroupNorm tile partials in its epilogue, then _NeckTail's two kernels.
    Backward: conv_p, then catre_op_rot_l1_bwd_lp - the GroupNorm sums from the forward's moments and ONE pass over (y, a)
    that rebuilds the linear's output gradient in LDS and takes da and dW from it on the bf16 matrix pipe (k_rot_l1_bwd_bf)
    instead of the apply pass + a dgrad and a wgrad launch that each re-read a [B*P,256] fp32 matrix."""

    @staticmethod
    def forward(ctx, a, w, b, gamma, beta, wn, bn, wp, bp, B, N, M, pre=None):
        lib = hip.load()
        ac, w2, wn = _c(a), _c(w.reshape(256, -1)), _c(wn)
        bc, bnc = _c(b), (_c(bn) if bn is not None else None)
        wv = _c(wp.reshape(-1))
        dev = a.device
        R, P = ac.shape[0], N + M
        if pre is not None:   # split mode: (y, partials) from the fused forward (rot_heads_forward); the backward is the
            y, part = pre     # one-pass kernel with hi + lo operands (k_rot_l1_bwd_sp)
            ctx.lp = "split"
        else:
            ctx.lp = True
            pk = _pack_bf16(w2, 256, 256, dev)
            y = torch.empty(R, 256, dtype=torch.float32, device=dev)
            part = torch.empty(R // 64, 32, 2, dtype=torch.float32, device=dev)
            hip.check(lib.catre_op_gemm_rows_gn(hip.ptr(ac), ac.stride(0), hip.ptr(pk), hip.ptr(bc), 0, hip.ptr(y), 256, 256,
                                                256, B, N, M, hip.ptr(part), 1, _st(a)), "catre_op_gemm_rows_gn")
        y3 = torch.empty(R, 3, dtype=torch.float32, device=dev)
        stat = torch.empty(B, 32, 2, dtype=torch.float32, device=dev)
        spart = torch.empty(R // 64, 3, 256, dtype=torch.float32, device=dev)
        hip.check(lib.catre_op_gnp_gelu_neck_fwd_s(hip.ptr(y), hip.ptr(part), hip.ptr(gamma), hip.ptr(beta), hip.ptr(wn),
                                                   hip.ptr(bnc), hip.ptr(wv), hip.ptr(y3), hip.ptr(stat), hip.ptr(spart), B, P,
                                                   _st(a)), "catre_op_gnp_gelu_neck_fwd_s")
        out = torch.empty(B, 3, dtype=torch.float32, device=dev)
        hip.check(lib.catre_op_wsum_fwd(hip.ptr(y3), hip.ptr(wv), hip.ptr(bp), hip.ptr(out), B, P, _st(a)), "catre_op_wsum_fwd")
        ctx.save_for_backward(ac, w2, y, stat, gamma, beta, wn, spart, y3, wv)
        ctx.dims, ctx.wshape, ctx.wp_shape = (B, P), w.shape, wp.shape
        ctx.has_bn, ctx.has_bp = bn is not None, bp is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        a, w2, y, stat, gamma, beta, wn, spart, y3, wv = ctx.saved_tensors
        B, P = ctx.dims
        lib = hip.load()
        dev = a.device
        dout = _c(dout)
        dy3, dwp, dbp, dbn = _wsum_backward(dout, y3, wv, B, P, ctx.has_bp, ctx.has_bn)
        da, dw, db, dpar = _rot_l1_backward(dy3, a, w2, y, stat, gamma, beta, wn, B, P, dout=dout, spart=spart, lp=ctx.lp)
        return (da, dw.view(ctx.wshape), db, dpar[0], dpar[1], dpar[2:5], dbn, dwp.view(ctx.wp_shape), dbp, None, None, None,
                None)


class _RotHeadLP(torch.autograd.Function):
    """A whole RotHead behind the per-cloud bias of its first layer under torch.autocast (conv_out_per_rot_head.py:126-140:
    Conv1d 64 -> 256 + per-cloud bias, GroupNorm, GELU, Conv1d 256 -> 256, GroupNorm, GELU, neck, conv_p) as ONE node:
    x [B*P,64] (pointfeat, object-major) -> out [B,3].  The three [B*P,256] activations the backward needs - y0, a0, y1 - and
    the gradient that travels between the two backward passes are bf16 rows (what autocast's Conv1d outputs are): every pass
    over them moves half the bytes of the fp32-row form (_RotL0Block + _RotL1TailLP), nothing else changes - bf16-operand
    GEMMs with fp32 accumulation, fp32 GroupNorm statistics from the GEMM epilogues, fp32 GroupNorm / GELU arithmetic.
    Forward: 2 x (row GEMM, GroupNorm + GELU pass), conv_p.  Backward: conv_p, k_rot_l1_bwd_bf, GroupNorm-0 sums,
    k_rot_l0_bwd_bf (include/catre_hip.h: the catre_op_*_h entry points)."""

    @staticmethod
    def forward(ctx, x, w0, bias2d, g0, be0, w1, b1, g1, be1, wn, bn, wp, bp, B, N, M, x_cm=False):
        out, saved, meta = _lp_head_forward(x, w0, bias2d, g0, be0, w1, b1, g1, be1, wn, bn, wp, bp, B, N, M, x_cm)
        ctx.save_for_backward(*saved)
        ctx.meta = meta
        return out

    @staticmethod
    def backward(ctx, dout):
        return _lp_head_backward(ctx.saved_tensors, ctx.meta, dout, None) + (None, None, None, None)


def _lp_head_forward(x, w0, bias2d, g0, be0, w1, b1, g1, be1, wn, bn, wp, bp, B, N, M, x_cm):
    """Forward of one autocast RotHead (:class:`_RotHeadLP`) -> (out [B,3], tensors to save, meta)."""
    lib = hip.load()
    dev = x.device
    xc, bc = _c(x), _c(bias2d)
    w0c, w1c, wn = _c(w0.reshape(256, -1)), _c(w1.reshape(256, -1)), _c(wn)
    b1c, bnc = _c(b1), (_c(bn) if bn is not None else None)
    wv = _c(wp.reshape(-1))
    R, P = xc.shape[0], N + M
    st = _st(x)
    h = lambda: torch.empty(R, 256, dtype=torch.bfloat16, device=dev)
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    y0, a0, y1 = h(), h(), h()
    part0, part1 = f(R // 64, 32, 2), f(R // 64, 32, 2)
    stat0, stat1 = f(B, 32, 2), f(B, 32, 2)
    pk0 = _pack_bf16(w0c, 256, 64, dev)
    # x_cm: x is pointfeat in the trunk kernel's cloud-major row order (no object-major copy was made); y0 and every
    # tensor behind it are object-major, and so is the gradient this node returns for x (train_ops._PointfeatHub)
    hip.check(lib.catre_op_gemm_rows_gn_h(hip.ptr(xc), xc.stride(0), hip.ptr(pk0), hip.ptr(bc), 2 if x_cm else 1, hip.ptr(y0),
                                          256, 256, 64, B, N, M, hip.ptr(part0), 2, st), "catre_op_gemm_rows_gn_h")
    pk1 = _pack_bf16(w1c, 256, 256, dev)
    if knobs().lp_rot_fuse_gn0:   # GroupNorm-0 + GELU inside the second linear's operand staging (same values, one launch less)
        hip.check(lib.catre_op_gn_gelu_gemm_rows_h(hip.ptr(y0), hip.ptr(part0), hip.ptr(g0), hip.ptr(be0), hip.ptr(a0),
                                                   hip.ptr(stat0), hip.ptr(pk1), hip.ptr(b1c), hip.ptr(y1), hip.ptr(part1),
                                                   B, N, M, st), "catre_op_gn_gelu_gemm_rows_h")
    else:
        hip.check(lib.catre_op_gnp_gelu_fwd_pre_h(hip.ptr(y0), hip.ptr(part0), hip.ptr(g0), hip.ptr(be0), hip.ptr(a0),
                                                  hip.ptr(stat0), B, P, st), "catre_op_gnp_gelu_fwd_pre_h")
        hip.check(lib.catre_op_gemm_rows_gn_h(hip.ptr(a0), 256, hip.ptr(pk1), hip.ptr(b1c), 0, hip.ptr(y1), 256, 256, 256,
                                              B, N, M, hip.ptr(part1), 3, st), "catre_op_gemm_rows_gn_h")
    y3, spart = f(R, 3), f(R // 64, 3, 256)
    hip.check(lib.catre_op_gnp_gelu_neck_fwd_s_h(hip.ptr(y1), hip.ptr(part1), hip.ptr(g1), hip.ptr(be1), hip.ptr(wn),
                                                 hip.ptr(bnc), hip.ptr(wv), hip.ptr(y3), hip.ptr(stat1), hip.ptr(spart), B, P,
                                                 st), "catre_op_gnp_gelu_neck_fwd_s_h")
    out = f(B, 3)
    hip.check(lib.catre_op_wsum_fwd(hip.ptr(y3), hip.ptr(wv), hip.ptr(bp), hip.ptr(out), B, P, st), "catre_op_wsum_fwd")
    saved = (xc, w0c, y0, stat0, g0, be0, a0, w1c, y1, stat1, g1, be1, wn, spart, y3, wv)
    meta = dict(dims=(B, N, M), x_cm=bool(x_cm), shapes=(w0.shape, w1.shape, wp.shape), has_bn=bn is not None,
                has_bp=bp is not None)
    return out, saved, meta


def _lp_head_backward(saved, meta, dout, dx_acc):
    """Backward of one autocast RotHead -> gradients of (x, w0, bias2d, g0, be0, w1, b1, g1, be1, wn, bn, wp, bp).  dx_acc: a
    [R,64] tensor the data gradient is ADDED to (the other head's: the pair node sums the two without a pass of its own)."""
    x, w0, y0, stat0, g0, be0, a0, w1, y1, stat1, g1, be1, wn, spart, y3, wv = saved
    B, N, M = meta["dims"]
    P = N + M
    lib = hip.load()
    dev = x.device
    dout = _c(dout)
    st = _st(dout)
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    dy3, dwp, dbp, dbn = _wsum_backward(dout, y3, wv, B, P, meta["has_bp"], meta["has_bn"])
    da0 = torch.empty(B * P, 256, dtype=torch.bfloat16, device=dev)
    dwb1, dpar1 = f(256 * 256 + 256), f(5, 256)
    ws = _ws(lib.catre_op_rot_l1_bwd_ws_bytes(B, P), dev)
    hip.check(lib.catre_op_rot_l1_bwd_h(hip.ptr(dy3), hip.ptr(dout), hip.ptr(spart), hip.ptr(y1), hip.ptr(stat1), hip.ptr(g1),
                                        hip.ptr(be1), hip.ptr(wn), hip.ptr(a0), hip.ptr(w1), hip.ptr(da0), hip.ptr(dwb1),
                                        hip.ptr(dpar1), hip.ptr(ws), ws.numel(), B, P, st), "catre_op_rot_l1_bwd_h")
    dx = dx_acc if dx_acc is not None else f(x.shape[0], 64)
    dw0 = f(256, 64)
    db0 = f(2 * B if M > 0 else B, 256)
    dg0, dbe0 = torch.empty_like(g0), torch.empty_like(be0)
    ws = _ws(lib.catre_op_rot_l0_bwd_ws_bytes(B, N, M), dev)
    hip.check(lib.catre_op_rot_l0_bwd_h(hip.ptr(da0), hip.ptr(y0), hip.ptr(stat0), hip.ptr(g0), hip.ptr(be0), hip.ptr(x),
                                        x.stride(0), hip.ptr(w0), hip.ptr(dx), 64, hip.ptr(dw0), hip.ptr(db0), hip.ptr(dg0),
                                        hip.ptr(dbe0), (1 if dx_acc is not None else 0) | (2 if meta["x_cm"] else 0),
                                        hip.ptr(ws), ws.numel(), B, N, M, st), "catre_op_rot_l0_bwd_h")
    s0, s1, sp = meta["shapes"]
    return (dx, dw0.view(s0), db0, dg0, dbe0, dwb1[: 256 * 256].view(s1), dwb1[256 * 256:], dpar1[0], dpar1[1], dpar1[2:5],
            dbn, dwp.view(sp), dbp)


class _RotHeadPairLP(torch.autograd.Function):
    """Both autocast RotHeads (:class:`_RotHeadLP` twice) as ONE node: the same kernels in the same order, but the second
    head's layer-0 backward ADDS its data gradient to the first one's (`catre_op_rot_l0_bwd_h`, accumulate flag) instead of
    autograd summing two [B*(N+M),64] tensors in a pass of its own (134 MB read twice and written once per iteration).
    Inputs: x, then the 12 per-head tensors of head x, then those of head y, B, N, M, x_cm -> (out_x [B,3], out_y [B,3])."""

    @staticmethod
    def forward(ctx, x, *a):
        hx, hy, (B, N, M, x_cm) = a[:12], a[12:24], a[24:]
        ox, sx, mx = _lp_head_forward(x, *hx, B, N, M, x_cm)
        oy, sy, my = _lp_head_forward(x, *hy, B, N, M, x_cm)
        ctx.save_for_backward(*sx, *sy)
        ctx.meta = (mx, my, len(sx))
        return ox, oy

    @staticmethod
    def backward(ctx, dox, doy):
        mx, my, n = ctx.meta
        sx, sy = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        if dox is None or doy is None:
            z = torch.zeros_like(dox if dox is not None else doy)
            dox, doy = (dox if dox is not None else z), (doy if doy is not None else z)
        gx = _lp_head_backward(sx, mx, dox, None)
        gy = _lp_head_backward(sy, my, doy, gx[0])           # dx: the two heads' sum, accumulated by the second kernel
        return (gy[0],) + gx[1:] + gy[1:] + (None, None, None, None)


def rot_head_pair_lp(x, head_x, head_y, B, N, M, x_cm=False):
    """head_x / head_y: (w0, bias2d, g0, be0, w1, b1, g1, be1, wn, bn, wp, bp) of the two RotHeads -> (out_x, out_y)."""
    return _RotHeadPairLP.apply(x, *head_x, *head_y, B, N, M, x_cm)


def rot_head_lp_ok(x, w0, w1, b1, N, M):
    return (knobs().fused_lp_rot and knobs().lp_rot_bf16_rows and _amp() == 1 and x.shape[1] == 64 and w0.shape[0] == 256
            and w0.reshape(256, -1).shape[1] == 64 and w1.shape[0] == 256 and w1.reshape(256, -1).shape[1] == 256
            and b1 is not None and N % 64 == 0 and M % 64 == 0 and N > 0 and M > 0)


def rot_head_lp(x, w0, bias2d, g0, be0, w1, b1, g1, be1, wn, bn, wp, bp, B, N, M, x_cm=False):
    """The RotHead behind its per-cloud layer-0 bias under autocast -> [B,3] (rot_head_lp_ok); x [B*(N+M),64] object-major,
    w0 [256,64] the point half of layers.0, bias2d [2B,256] its global half + bias, wn [3,256], bn [3] or None.  x_cm: the rows
    of x are cloud-major (pointfeat as the trunk kernel wrote it); the gradient returned for x stays object-major."""
    return _RotHeadLP.apply(x, w0, bias2d, g0, be0, w1, b1, g1, be1, wn, bn, wp, bp, B, N, M, x_cm)


def rot_l1_tail_lp_ok(a, w, b, N, M):
    return (knobs().fused_lp_rot and _amp() == 1 and a.shape[1] == 256 and w.shape[0] == 256 and w.reshape(256, -1).shape[1] == 256 and b is not None
            and N % 64 == 0 and M % 64 == 0 and N > 0 and M > 0)


def rot_l1_tail_lp(a, w, b, gamma, beta, wn, bn, wp, bp, B, N, M, pre=None):
    """conv_p(neck(gelu(GroupNorm(a w^T + b)))) -> [B,3] under autocast (rot_l1_tail_lp_ok); a [B*(N+M),256] object-major.
    pre = (y, partials) from the split mode's fused forward: graph node only, split one-pass backward."""
    return _RotL1TailLP.apply(a, w, b, gamma, beta, wn, bn, wp, bp, B, N, M, pre)


class _RotL0Block(torch.autograd.Function):
    """A RotHead's first block (fp32, or bf16-operand GEMMs under autocast) - 64 -> 256 linear with a per-cloud bias, GroupNorm(32,256), GELU
    (conv_out_per_rot_head.py:126-131) - as one graph node.  Forward: the two kernels of linear_cloudbias +
    gn_points_gelu.  Backward: the GroupNorm sums, then ONE pass over (da, y) that rebuilds the linear's output gradient
    tile by tile in LDS and takes dx, dW and the per-cloud bias gradient from it (catre_op_rot_l0_bwd) - instead of
    materialising that [R,256] gradient and reading it three more times."""

    @staticmethod
    def forward(ctx, x, w, bias2d, gamma, beta, B, N, M, pre=None):
        lib = hip.load()
        xc, w2, bc = _c(x), _c(w.reshape(256, -1)), _c(bias2d)
        R, P = xc.shape[0], N + M
        if pre is not None:   # (y, a, stat) from a fused forward (rot_heads_forward, split mode): only the graph node; its
            y, a, stat = pre  # backward is the one-pass kernel with hi + lo operands (k_rot_l0_bwd_sp): fp32-grade
            ctx.save_for_backward(xc, w2, y, stat, gamma, beta)
            ctx.dims, ctx.wshape, ctx.amp = (B, N, M), w.shape, (2 if knobs().split_l0_sp else 0)
            return a
        amp = 1 if _amp() == 1 else 0   # autocast: the forward linear on bf16 operands (what _RotLinear does there)
        if amp:
            wp = _pack_bf16(w2, 256, 64, x.device)
        else:
            wp = torch.empty(256 * 64, dtype=torch.float32, device=x.device)
            hip.check(lib.catre_op_pack(hip.ptr(w2), w2.stride(0), 256, 64, 0, hip.ptr(wp), _st(x)), "catre_op_pack")
        y = torch.empty(R, 256, dtype=torch.float32, device=x.device)
        part = torch.empty(R // 64, 32, 2, dtype=torch.float32, device=x.device)
        hip.check(lib.catre_op_gemm_rows_gn(hip.ptr(xc), xc.stride(0), hip.ptr(wp), hip.ptr(bc), 1, hip.ptr(y), 256, 256, 64,
                                            B, N, M, hip.ptr(part), amp, _st(x)), "catre_op_gemm_rows_gn")
        a = torch.empty_like(y)
        stat = torch.empty(B, 32, 2, dtype=torch.float32, device=x.device)
        hip.check(lib.catre_op_gnp_gelu_fwd_pre(hip.ptr(y), hip.ptr(part), hip.ptr(gamma), hip.ptr(beta), hip.ptr(a),
                                                hip.ptr(stat), B, P, _st(x)), "catre_op_gnp_gelu_fwd_pre")
        ctx.save_for_backward(xc, w2, y, stat, gamma, beta)
        ctx.dims, ctx.wshape, ctx.amp = (B, N, M), w.shape, amp
        return a

    @staticmethod
    def backward(ctx, da):
        x, w2, y, stat, gamma, beta = ctx.saved_tensors
        B, N, M = ctx.dims
        lib = hip.load()
        da = _c(da)
        dev = da.device
        dx = torch.empty(x.shape[0], 64, dtype=torch.float32, device=dev)
        dw = torch.empty(256, 64, dtype=torch.float32, device=dev)
        db = torch.empty(2 * B if M > 0 else B, 256, dtype=torch.float32, device=dev)
        dg, dbe = torch.empty_like(gamma), torch.empty_like(beta)
        ws = _ws(lib.catre_op_rot_l0_bwd_ws_bytes(B, N, M), dev)
        # autocast: both GEMMs on the bf16 pipe; split: hi + lo operands
        fn = (lib.catre_op_rot_l0_bwd_sp if ctx.amp == 2 else lib.catre_op_rot_l0_bwd_lp) if ctx.amp else lib.catre_op_rot_l0_bwd
        hip.check(fn(hip.ptr(da), hip.ptr(y), hip.ptr(stat), hip.ptr(gamma), hip.ptr(beta), hip.ptr(x),
                     x.stride(0), hip.ptr(w2), hip.ptr(dx), 64, hip.ptr(dw), hip.ptr(db), hip.ptr(dg),
                     hip.ptr(dbe), 0, hip.ptr(ws), ws.numel(), B, N, M, _st(da)), "catre_op_rot_l0_bwd")
        return dx, dw.view(ctx.wshape), db, dg, dbe, None, None, None, None


def rot_l0_block_ok(x, w, N, M):
    # autocast takes the node too (knob fused_lp_rot): bf16-operand forward, the one-pass backward
    return ((_amp() == 0 or (_amp() == 1 and knobs().fused_lp_rot)) and w.shape[0] == 256 and w.reshape(256, -1).shape[1] == 64 and x.shape[1] == 64
            and N % 64 == 0 and M % 64 == 0 and N > 0)


def rot_l0_block(x, w, bias2d, gamma, beta, B, N, M, pre=None):
    """gelu(GroupNorm(x w^T + bias2d[cloud])) for x [B*(N+M),64] object-major, w [256,64], bias2d [2B,256] (fp32 mode or
    autocast, N and M multiples of 64: rot_l0_block_ok).  pre: (y, a, stat) already computed by a fused forward."""
    return _RotL0Block.apply(x, w, bias2d, gamma, beta, B, N, M, pre)


class _RotL1Block(torch.autograd.Function):
    """A RotHead's second block in fp32 - 256 -> 256 linear, GroupNorm(32,256), GELU, neck conv (conv_out_per_rot_head.py:
    129-137) - as one graph node: y3 [B*P,3] from the block's input a [B*P,256].  Forward: the linear with GroupNorm tile
    partials in its epilogue, then gn_points_gelu_neck's kernel.  Backward: the neck / GroupNorm sums, then ONE pass over
    (y, a) that keeps the linear's output gradient in LDS and takes da and dW from it (catre_op_rot_l1_bwd)."""

    @staticmethod
    def forward(ctx, a, w, b, gamma, beta, wn, bn, B, N, M):
        lib = hip.load()
        ac, w2, wn = _c(a), _c(w.reshape(256, -1)), _c(wn)
        bc, bnc = _c(b), (_c(bn) if bn is not None else None)
        R, P = ac.shape[0], N + M
        wp = torch.empty(256 * 256, dtype=torch.float32, device=a.device)
        hip.check(lib.catre_op_pack(hip.ptr(w2), w2.stride(0), 256, 256, 0, hip.ptr(wp), _st(a)), "catre_op_pack")
        y = torch.empty(R, 256, dtype=torch.float32, device=a.device)
        part = torch.empty(R // 64, 32, 2, dtype=torch.float32, device=a.device)
        hip.check(lib.catre_op_gemm_rows_gn(hip.ptr(ac), ac.stride(0), hip.ptr(wp), hip.ptr(bc), 0, hip.ptr(y), 256, 256, 256,
                                            B, N, M, hip.ptr(part), 0, _st(a)), "catre_op_gemm_rows_gn")
        y3 = torch.empty(R, 3, dtype=torch.float32, device=a.device)
        stat = torch.empty(B, 32, 2, dtype=torch.float32, device=a.device)
        hip.check(lib.catre_op_gnp_gelu_neck_fwd(hip.ptr(y), hip.ptr(part), hip.ptr(gamma), hip.ptr(beta), hip.ptr(wn),
                                                 hip.ptr(bnc), hip.ptr(y3), hip.ptr(stat), B, P, _st(a)),
                  "catre_op_gnp_gelu_neck_fwd")
        ctx.save_for_backward(ac, w2, y, stat, gamma, beta, wn)
        ctx.dims, ctx.wshape, ctx.has_bn = (B, P), w.shape, bn is not None
        return y3

    @staticmethod
    def backward(ctx, dy3):
        a, w2, y, stat, gamma, beta, wn = ctx.saved_tensors
        B, P = ctx.dims
        lib = hip.load()
        dy3 = _c(dy3)
        dev = dy3.device
        da = torch.empty_like(a)
        dwb = torch.empty(256 * 256 + 256, dtype=torch.float32, device=dev)
        dpar = torch.empty(5, 256, dtype=torch.float32, device=dev)
        ws = _ws(lib.catre_op_rot_l1_bwd_ws_bytes(B, P), dev)
        hip.check(lib.catre_op_rot_l1_bwd(hip.ptr(dy3), hip.ptr(y), hip.ptr(stat), hip.ptr(gamma), hip.ptr(beta), hip.ptr(wn),
                                          hip.ptr(a), hip.ptr(w2), hip.ptr(da), hip.ptr(dwb), hip.ptr(dpar), hip.ptr(ws),
                                          ws.numel(), B, P, _st(dy3)), "catre_op_rot_l1_bwd")
        dbn = _colsum(dy3) if ctx.has_bn else None
        return (da, dwb[: 256 * 256].view(ctx.wshape), dwb[256 * 256:], dpar[0], dpar[1], dpar[2:5], dbn, None, None, None)


def rot_l1_block_ok(a, w, N, M):
    return (_amp() == 0 and a.shape[1] == 256 and w.shape[0] == 256 and w.reshape(256, -1).shape[1] == 256
            and N % 64 == 0 and M % 64 == 0 and N > 0)


def rot_l1_block(a, w, b, gamma, beta, wn, bn, B, N, M):
    """neck(gelu(GroupNorm(a w^T + b))) -> [B*(N+M), 3] for a [B*(N+M),256] object-major, w [256,256], wn [3,256] (fp32 mode,
    N and M multiples of 64: rot_l1_block_ok)."""
    return _RotL1Block.apply(a, w, b, gamma, beta, wn, bn, B, N, M)


def _rot_l1_backward(dy3, a, w2, y, stat, gamma, beta, wn, B, P, dout=None, spart=None, lp=False):
    """_RotL1Block's backward on explicit tensors -> (da, dW [256,256], db [256], dpar [5,256]).  With (dout [B,3], spart):
    the GroupNorm sums come from the forward's tile moments (catre_op_gnp_gelu_neck_fwd_s) instead of a pass over y.
    lp: the two GEMMs on the bf16 matrix pipe (autocast)."""
    lib = hip.load()
    dy3 = _c(dy3)
    dev = dy3.device
    da = torch.empty_like(a)
    dwb = torch.empty(256 * 256 + 256, dtype=torch.float32, device=dev)
    dpar = torch.empty(5, 256, dtype=torch.float32, device=dev)
    ws = _ws(lib.catre_op_rot_l1_bwd_ws_bytes(B, P), dev)
    if lp:
        fn = lib.catre_op_rot_l1_bwd_sp if lp == "split" else lib.catre_op_rot_l1_bwd_lp
        hip.check(fn(hip.ptr(dy3), hip.ptr(dout), hip.ptr(spart), hip.ptr(y), hip.ptr(stat),
                                             hip.ptr(gamma), hip.ptr(beta), hip.ptr(wn), hip.ptr(a), hip.ptr(w2), hip.ptr(da),
                                             hip.ptr(dwb), hip.ptr(dpar), hip.ptr(ws), ws.numel(), B, P, _st(dy3)),
                  "catre_op_rot_l1_bwd_lp")
    elif spart is not None:
        hip.check(lib.catre_op_rot_l1_bwd_s(hip.ptr(dy3), hip.ptr(dout), hip.ptr(spart), hip.ptr(y), hip.ptr(stat),
                                            hip.ptr(gamma), hip.ptr(beta), hip.ptr(wn), hip.ptr(a), hip.ptr(w2), hip.ptr(da),
                                            hip.ptr(dwb), hip.ptr(dpar), hip.ptr(ws), ws.numel(), B, P, _st(dy3)),
                  "catre_op_rot_l1_bwd_s")
    else:
        hip.check(lib.catre_op_rot_l1_bwd(hip.ptr(dy3), hip.ptr(y), hip.ptr(stat), hip.ptr(gamma), hip.ptr(beta), hip.ptr(wn),
                                          hip.ptr(a), hip.ptr(w2), hip.ptr(da), hip.ptr(dwb), hip.ptr(dpar), hip.ptr(ws),
                                          ws.numel(), B, P, _st(dy3)), "catre_op_rot_l1_bwd")
    return da, dwb[: 256 * 256].view(256, 256), dwb[256 * 256:], dpar


def _rot_l0_backward(da, x, w2, y, stat, gamma, beta, B, N, M, dx_acc=None, x_cloud_major=False):
    """_RotL0Block's backward on explicit tensors -> (dx [R,64], dW [256,64], dbias2d [2B,256], dgamma, dbeta).  dx_acc: a
    [R,64] tensor the data gradient is ADDED to (and returned) instead of a fresh one.  x_cloud_major: the rows of x are
    cloud-major (dx stays object-major)."""
    lib = hip.load()
    da = _c(da)
    dev = da.device
    dx = dx_acc if dx_acc is not None else torch.empty(x.shape[0], 64, dtype=torch.float32, device=dev)
    dw = torch.empty(256, 64, dtype=torch.float32, device=dev)
    db = torch.empty(2 * B if M > 0 else B, 256, dtype=torch.float32, device=dev)
    dg, dbe = torch.empty_like(gamma), torch.empty_like(beta)
    ws = _ws(lib.catre_op_rot_l0_bwd_ws_bytes(B, N, M), dev)
    hip.check(lib.catre_op_rot_l0_bwd(hip.ptr(da), hip.ptr(y), hip.ptr(stat), hip.ptr(gamma), hip.ptr(beta), hip.ptr(x),
                                      x.stride(0), hip.ptr(w2), hip.ptr(dx), 64, hip.ptr(dw), hip.ptr(db), hip.ptr(dg),
                                      hip.ptr(dbe), int(dx_acc is not None) | (2 if x_cloud_major else 0), hip.ptr(ws),
                                      ws.numel(), B, N, M, _st(da)),
              "catre_op_rot_l0_bwd")
    return dx, dw, db, dg, dbe


class _RotHeads(torch.autograd.Function):
    """BOTH RotHeads up to the neck output (conv_out_per_rot_head.py:126-137) as one graph node in fp32.  Forward on the
    fused inference kernels with saves (``catre_train_rot_fwd``: GroupNorm-0 statistics from second moments of pointfeat,
    then layer 0 + GN0 + GELU + layer 1 per 64-point tile for both heads - no pass re-reads a [rows,256] matrix between the
    two linears), then each head's GroupNorm-1 + GELU + neck kernel and conv_p (the weighted sum over the points).  Because
    the node ends behind conv_p, every row's neck gradient is wp[p] * dout[b]: the neck kernel also leaves three per-channel
    moments per tile from which the backward gets the GroupNorm-1 sums / dgamma / dbeta / dWn without its reduction pass
    over y1 (catre_op_gnp_gelu_neck_fwd_s, catre_op_rot_l1_bwd_s).  Backward: conv_p, then the per-head passes of
    _RotL1Block and _RotL0Block on the saved head slices; the two pointfeat gradients are summed here.

    Per head h in (x, y) the inputs are: bias0 [2B,256] (global half of layer 0 + conv bias, per cloud), w0 [256,64] (local
    half), GN0 gamma/beta, w1 [256,256], b1, GN1 gamma/beta, wn [3,256], bn [3] or None, conv_p weight [1,P,1] and bias.  ``pf_cm`` is pointfeat cloud-major
    (what the kernels read); ``pf_obj`` the same rows object-major - the differentiable input and what the backward reads.
    ``prm`` / ``packed``: the runtime's parameter pointer array and fp32 weight image (heads included)."""

    NH = 12  # tensors per head

    @staticmethod
    def forward(ctx, pf_cm, pf_obj, prm, packed, B, N, M, *heads):
        lib = hip.load()
        hx, hy = heads[: _RotHeads.NH], heads[_RotHeads.NH:]
        # pf_obj on pf_cm's own buffer: the hub handed out pointfeat itself (cloud-major rows) instead of an object-major
        # copy - only the backward reads it (k_rot_l0_bwd's X operand), and it can walk either row order
        ctx.x_cm = pf_obj.data_ptr() == pf_cm.data_ptr()
        dev = pf_obj.device
        R, P = B * (N + M), N + M
        pf_cm, pf_obj = _c(pf_cm), _c(pf_obj)
        bias0 = torch.stack([_c(hx[0]), _c(hy[0])])                               # [2, 2B, 256]
        y0 = torch.empty(2, R, 256, dtype=torch.float32, device=dev)
        a0, y1 = torch.empty_like(y0), torch.empty_like(y0)
        part = torch.empty(2, R // 64, 32, 2, dtype=torch.float32, device=dev)
        stat0 = torch.empty(2, B, 32, 2, dtype=torch.float32, device=dev)
        ws = _ws(lib.catre_train_rot_fwd_ws_bytes(B), dev)
        hip.check(lib.catre_train_rot_fwd(hip.ptr(pf_cm), hip.ptr(bias0), prm, hip.ptr(packed), hip.ptr(y0), hip.ptr(a0),
                                          hip.ptr(y1), hip.ptr(part), hip.ptr(stat0), hip.ptr(ws), ws.numel(), B, N, M, 0,
                                          _st(pf_obj)), "catre_train_rot_fwd")
        stat1 = torch.empty(2, B, 32, 2, dtype=torch.float32, device=dev)
        outs, keep = [], []
        spart = torch.empty(2, R // 64, 3, 256, dtype=torch.float32, device=dev)
        y3 = torch.empty(2, R, 3, dtype=torch.float32, device=dev)
        for h, hd in enumerate((hx, hy)):
            _, w0, g0, be0, w1, b1, g1, be1, wn, bn, wp, bp = hd
            wn, wv = _c(wn), _c(wp.reshape(-1))
            bnc = _c(bn) if bn is not None else None
            hip.check(lib.catre_op_gnp_gelu_neck_fwd_s(hip.ptr(y1[h]), hip.ptr(part[h]), hip.ptr(g1), hip.ptr(be1),
                                                       hip.ptr(wn), hip.ptr(bnc), hip.ptr(wv), hip.ptr(y3[h]),
                                                       hip.ptr(stat1[h]), hip.ptr(spart[h]), B, P, _st(pf_obj)),
                      "catre_op_gnp_gelu_neck_fwd_s")
            out = torch.empty(B, 3, dtype=torch.float32, device=dev)
            hip.check(lib.catre_op_wsum_fwd(hip.ptr(y3[h]), hip.ptr(wv), hip.ptr(bp), hip.ptr(out), B, P, _st(pf_obj)),
                      "catre_op_wsum_fwd")
            outs.append(out)
            keep += [_c(w0.reshape(256, -1)), g0, be0, _c(w1.reshape(256, -1)), g1, be1, wn, wv]
        ctx.save_for_backward(pf_obj, y0, a0, y1, stat0, stat1, spart, y3, *keep)
        ctx.dims = (B, N, M)
        ctx.has_bn = (hx[9] is not None, hy[9] is not None)
        ctx.has_bp = (hx[11] is not None, hy[11] is not None)
        ctx.wshapes = (hx[1].shape, hx[4].shape, hy[1].shape, hy[4].shape, hx[10].shape, hy[10].shape)
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, doutx, douty):
        pf_obj, y0, a0, y1, stat0, stat1, spart, y3, *keep = ctx.saved_tensors
        B, N, M = ctx.dims
        P = N + M
        lib = hip.load()
        dev = pf_obj.device
        grads, dxs = [], []
        for h, dout in enumerate((doutx, douty)):
            w0, g0, be0, w1, g1, be1, wn, wv = keep[8 * h: 8 * h + 8]
            dout = _c(dout)
            # conv_p backward: dy3[b,p,:] = wp[p] dout[b,:], dwp, dbias (train_ops._WSum)
            dy3, dwp, dbp, dbn = _wsum_backward(dout, y3[h], wv, B, P, ctx.has_bp[h], ctx.has_bn[h])
            da, dw1, db1, dpar = _rot_l1_backward(dy3, a0[h], w1, y1[h], stat1[h], g1, be1, wn, B, P, dout=dout,
                                                  spart=spart[h])
            dx, dw0, dbias0, dg0, dbe0 = _rot_l0_backward(da, pf_obj, w0, y0[h], stat0[h], g0, be0, B, N, M,
                                                          dx_acc=dxs[0] if dxs else None,   # the second head adds into the first's dx
                                                          x_cloud_major=ctx.x_cm)
            del da
            dxs.append(dx)
            grads += [dbias0, dw0.view(ctx.wshapes[2 * h]), dg0, dbe0, dw1.view(ctx.wshapes[2 * h + 1]), db1, dpar[0], dpar[1],
                      dpar[2:5], dbn, dwp.view(ctx.wshapes[4 + h]), dbp]
        return (None, dxs[0], None, None, None, None, None) + tuple(grads)


def rot_heads_forward(pf_cm, bias0x, bias0y, prm, packed, B, N, M, mode):
    """The fused forward of both RotHeads up to the GroupNorm-1 input WITHOUT graph nodes (`catre_train_rot_fwd` in
    `mode`: 0 fp32, 2 split) -> dict of head-major buffers y0, a0, y1 [2,R,256], part1 [2,R/64,32,2], stat0 [2,B,32,2] for
    the per-head ops' `pre=` arguments (the split mode's graph: its backward stays the layer-wise split dgrad / wgrad)."""
    lib = hip.load()
    dev = pf_cm.device
    R = B * (N + M)
    pf_cm = _c(pf_cm)
    bias0 = torch.stack([_c(bias0x.detach()), _c(bias0y.detach())])
    y0 = torch.empty(2, R, 256, dtype=torch.float32, device=dev)
    a0, y1 = torch.empty_like(y0), torch.empty_like(y0)
    part = torch.empty(2, R // 64, 32, 2, dtype=torch.float32, device=dev)
    stat0 = torch.empty(2, B, 32, 2, dtype=torch.float32, device=dev)
    ws = _ws(lib.catre_train_rot_fwd_ws_bytes(B), dev)
    hip.check(lib.catre_train_rot_fwd(hip.ptr(pf_cm), hip.ptr(bias0), prm, hip.ptr(packed), hip.ptr(y0), hip.ptr(a0),
                                      hip.ptr(y1), hip.ptr(part), hip.ptr(stat0), hip.ptr(ws), ws.numel(), B, N, M, int(mode),
                                      _st(pf_cm)), "catre_train_rot_fwd")
    return dict(y0=y0, a0=a0, y1=y1, part1=part, stat0=stat0)


def rot_heads_ok(pf_obj, N, M):
    return _amp() == 0 and pf_obj.shape[1] == 64 and N % 64 == 0 and M % 64 == 0 and N > 0 and M > 0


def rot_heads(pf_cm, pf_obj, prm, packed, B, N, M, head_x, head_y):
    """head_* = (bias0, w0_local, gn0_w, gn0_b, w1, b1, gn1_w, gn1_b, wn3, bn3, conv_p_w, conv_p_b) -> the two heads'
    conv_p outputs, each [B, 3] (columns >= rot_dim are zero)."""
    assert len(head_x) == _RotHeads.NH and len(head_y) == _RotHeads.NH
    return _RotHeads.apply(pf_cm, pf_obj, prm, packed, B, N, M, *head_x, *head_y)


class _GNRowsGelu(torch.autograd.Function):
    """gelu(GroupNorm(32,256)(y)) on a [R,256] matrix (groups of 8 channels inside each row; ts head)."""

    @staticmethod
    def forward(ctx, y, gamma, beta):
        lib = hip.load()
        y = _c(y)
        a = torch.empty_like(y)
        hip.check(lib.catre_op_gnr_gelu_fwd(hip.ptr(y), hip.ptr(gamma), hip.ptr(beta), hip.ptr(a), y.shape[0], _st(y)),
                  "catre_op_gnr_gelu_fwd")
        ctx.save_for_backward(y, gamma, beta)
        return a

    @staticmethod
    def backward(ctx, da):
        y, gamma, beta = ctx.saved_tensors
        lib = hip.load()
        da = _c(da)
        R = y.shape[0]
        dy = torch.empty_like(y)
        dg, db = torch.empty_like(gamma), torch.empty_like(beta)
        ws = _ws(R * 512 * 4, y.device)
        hip.check(lib.catre_op_gnr_gelu_bwd(hip.ptr(da), hip.ptr(y), hip.ptr(gamma), hip.ptr(beta), hip.ptr(dy),
                                            hip.ptr(dg), hip.ptr(db), 0, hip.ptr(ws), ws.numel(), R, _st(y)),
                  "catre_op_gnr_gelu_bwd")
        return dy, dg, db


def gn_rows_gelu(y, gamma, beta):
    return _GNRowsGelu.apply(y, gamma, beta)


# ------------------------------------------------------------------------------------------------- conv_p
class _WSum(torch.autograd.Function):
    """out[b,:] = sum_p w[p] * y[b*P+p, :3] + bias  (the point-wise Conv1d(P,1,1) of RotHead)."""

    @staticmethod
    def forward(ctx, y3, w, bias, B, P):
        lib = hip.load()
        y3 = _c(y3)
        wv = _c(w.reshape(-1))
        out = torch.empty(B, 3, dtype=torch.float32, device=y3.device)
        hip.check(lib.catre_op_wsum_fwd(hip.ptr(y3), hip.ptr(wv), hip.ptr(bias), hip.ptr(out), B, P, _st(y3)), "catre_op_wsum_fwd")
        ctx.save_for_backward(y3, w, bias)
        ctx.dims = (B, P)
        return out

    @staticmethod
    def backward(ctx, dout):
        y3, w, bias = ctx.saved_tensors
        B, P = ctx.dims
        lib = hip.load()
        dout = _c(dout)
        wv = _c(w.reshape(-1))
        dy = torch.empty_like(y3)
        dw = torch.empty(P, dtype=torch.float32, device=y3.device)
        dbias = torch.empty(1, dtype=torch.float32, device=y3.device) if bias is not None else None
        ws = _ws(B * P * 4, y3.device)
        hip.check(lib.catre_op_wsum_bwd(hip.ptr(dout), hip.ptr(y3), hip.ptr(wv), hip.ptr(dy), hip.ptr(dw), hip.ptr(dbias), 0,
                                        hip.ptr(ws), ws.numel(), B, P, _st(y3)), "catre_op_wsum_bwd")
        return dy, dw.reshape(w.shape), dbias, None, None


def weighted_point_sum(y3, w, bias, B, P):
    return _WSum.apply(y3, w, bias, B, P)


# ------------------------------------------------------------------------------------------------- pose update
class _PoseUpdate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rot6d, dt, ds, init_pose, init_scale, mean_scales, Ks, opts):
        from .runtime import pose_update

        pose, scale = pose_update(rot6d, dt, ds, init_pose, init_scale, mean_scales, Ks, opts)
        ctx.save_for_backward(rot6d, dt, ds, init_pose, init_scale, mean_scales, Ks)
        ctx.opts = opts
        return pose, scale

    @staticmethod
    def backward(ctx, d_pose, d_scale):
        rot6d, dt, ds, init_pose, init_scale, mean_scales, Ks = ctx.saved_tensors
        lib = hip.load()
        B = rot6d.shape[0]
        g6, gt, gs = torch.empty_like(rot6d), torch.empty_like(dt), torch.empty_like(ds)
        # contiguous copies are bound to names: a temporary would be returned to the caching allocator (and its block
        # handed to the next copy) before the kernel that reads it is even enqueued
        keep = [_c(t) if t is not None else None
                for t in (d_pose, d_scale, rot6d, dt, ds, init_pose, init_scale, mean_scales, Ks)]
        hip.check(lib.catre_op_pose_update_bwd(*[hip.ptr(t) for t in keep], ctypes.byref(ctx.opts), hip.ptr(g6),
                                               hip.ptr(gt), hip.ptr(gs), B, _st(rot6d)), "catre_op_pose_update_bwd")
        del keep
        return g6, gt, gs, None, None, None, None, None


def pose_update_autograd(rot6d, dt, ds, init_pose, init_scale, mean_scales, Ks, opts):
    return _PoseUpdate.apply(_c(rot6d), _c(dt), _c(ds), _c(init_pose), _c(init_scale), mean_scales, Ks, opts)
