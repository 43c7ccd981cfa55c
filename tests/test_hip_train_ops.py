"""Each training op (forward AND backward HIP kernels, catre_op_*) against plain torch autograd in fp64 on the
same inputs.  Tolerance: fp32 re-association only (2e-5 abs + 2e-5 rel on O(1) data)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cmp(got, want, what, atol=2e-5, rtol=2e-5):
    np.testing.assert_allclose(got.detach().cpu().double().numpy(), want.detach().double().numpy(), atol=atol, rtol=rtol,
                               err_msg=what)


def _leaf(t):
    return t.clone().to(DEV).requires_grad_(True), t.clone().double().requires_grad_(True)


def _gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("R,K,J,relu", [(300, 64, 128, True), (700, 128, 512, True), (1000, 512, 1024, False),
                                        (520, 3, 64, True), (515, 256, 3, False), (6, 1091, 256, False),
                                        (10, 256, 9, False), (300, 256, 256, False), (257, 64, 64, True),
                                        # >= 2048 rows: the tiled MFMA row kernels (ragged last tile included); below, the
                                        # split-K linear
                                        (2100, 64, 128, True), (2300, 128, 512, True), (2112, 512, 1024, False),
                                        (2049, 256, 256, False), (2500, 256, 64, True), (2050, 8, 64, True)])
@pytest.mark.parametrize("mode", ["fp32", "split"])
def test_linear_fwd_bwd(R, K, J, relu, mode):
    """mode 'split': the tiled shapes run hi + lo bf16 operands with three products on the bf16 pipe: 16-17 bits of
    mantissa per product, i.e. ~1e-5 absolute on O(1) results (tolerance 1e-4; fp32 kernels 2e-5)."""
    from catre_amd import train_ops as T

    g = _gen(R + K + J)
    x, xr = _leaf(torch.randn(R, K, generator=g))
    w, wr = _leaf(torch.randn(J, K, 1, generator=g) / K ** 0.5)
    b, br = _leaf(torch.randn(J, generator=g) * 0.1)
    with T.amp_mode(mode):
        y = T.linear(x, w, b, relu=relu)
    yr = F.linear(xr, wr[:, :, 0], br)
    # the reference takes the ReLU mask the device used: an output within the rounding error of zero (a few per 10^5
    # at split precision) may sit on either side, and one flipped mask entry changes a whole row of dx
    yr = yr * (y.detach() > 0).double().cpu() if relu else yr
    tol = dict(atol=1e-4, rtol=2e-5) if mode == "split" else {}
    _cmp(y, yr, "y", **tol)
    dy = torch.randn(R, J, generator=g)
    y.backward(dy.to(DEV))
    yr.backward(dy.double())
    _cmp(x.grad, xr.grad, "dx", **tol)
    _cmp(w.grad, wr.grad, "dw", atol=2e-3 if mode == "split" else 2e-4, rtol=2e-5)
    _cmp(b.grad, br.grad, "db", atol=2e-4, rtol=2e-5)


@pytest.mark.parametrize("R,cols,Kw,relu,two", [(524, 8, 3, True, True), (64, 8, 3, True, False), (5000, 4, 4, False, True),
                                               (33, 8, 1, True, True), (70000, 8, 3, True, True)])
@pytest.mark.parametrize("mode", ["fp32", "split"])
def test_two_consumer_point_layer_backward_in_one_pass(R, cols, Kw, relu, two, mode):
    """h1 = relu(conv1(x1)) with its two consumers (pointnet.py:103-109): linear_fan2 hands the buffer out twice and its
    backward (catre_op_skinny_bwd) adds the two gradient streams, masks, and produces dx (zero in the padding columns),
    dW and db in one pass - against fp64 autograd of the same graph; fp32 FMAs in every compute mode."""
    from catre_amd import train_ops as T

    g = _gen(R + cols + Kw)
    x0 = torch.randn(R, cols, generator=g)
    x0[:, Kw:] = 0
    x, xr = _leaf(x0)
    w, wr = _leaf(torch.randn(64, Kw, 1, generator=g))
    b, br = _leaf(torch.randn(64, generator=g) * 0.1)
    with T.amp_mode(mode):
        ya, yb = T.linear_fan2(x, w, b, relu=relu)
    assert ya.data_ptr() == yb.data_ptr()
    yr = F.linear(xr[:, :Kw], wr[:, :, 0], br)
    yr = yr * (ya.detach() > 0).double().cpu() if relu else yr
    _cmp(ya, yr, "y")
    da, db_ = torch.randn(R, 64, generator=g), torch.randn(R, 64, generator=g)
    loss = (ya * da.to(DEV)).sum() + ((yb * db_.to(DEV)).sum() if two else 0)
    loss.backward()
    ((yr * da.double()).sum() + ((yr * db_.double()).sum() if two else 0)).backward()
    assert float(x.grad[:, Kw:].abs().max()) == 0 if Kw < cols else True
    _cmp(x.grad[:, :Kw], xr.grad[:, :Kw], "dx", atol=1e-4)
    _cmp(w.grad, wr.grad, "dw", atol=2e-3 if R > 10000 else 2e-4, rtol=2e-5)
    _cmp(b.grad, br.grad, "db", atol=2e-3 if R > 10000 else 2e-4, rtol=2e-5)


def test_row_list_ops_gather_scatter_and_indexed_gemms():
    """The live-row list of the row-sparse backward: catre_op_gather_rows / scatter_rows against torch indexing, and the
    indexed GEMMs (catre_op_gemm_tn_bias_nr reads X through the list, catre_op_gemm_rows_nr masks its output through it)
    against the same GEMMs on gathered copies - bit for bit (same kernels, same order)."""
    from catre_amd import train_ops as T

    g = _gen(77)
    R, K, J = 4096, 64, 128
    live = torch.rand(R, generator=g) < 0.3
    rows_h = torch.nonzero(live).flatten().to(torch.int32)
    n = rows_h.numel()
    rows = torch.zeros(R, dtype=torch.int32)
    rows[:n] = rows_h
    rows, count = rows.to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV)
    x = torch.randn(R, K, generator=g).to(DEV)
    y1 = torch.randn(R, K, generator=g).to(DEV)           # a dense "ReLU output": its sign pattern is the mask
    dy = torch.randn(R, J, generator=g).to(DEV)            # compact rows: only the first n are meaningful
    w = (torch.randn(J, K, generator=g) / J ** 0.5).to(DEV)
    xc = T._gather_rows(x, rows, count)
    assert torch.equal(xc[:n].cpu(), x.cpu()[rows_h.long()])
    for mode in ("fp32", "bf16", "split"):
        amp = T._MODES[mode]
        dw_a, db_a = T._wgrad_n(dy, xc, None, count, amp)
        dw_b, db_b = T._wgrad_n(dy, x, None, count, amp, x_rows=rows)
        assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b), mode
        y1c = T._gather_rows(y1, rows, count)
        dx_a = T._dgrad_n(dy, w, None, count, amp)
        dx_a = torch.where(y1c > 0, dx_a, torch.zeros_like(dx_a))
        dx_b = T._dgrad_n(dy, w, None, count, amp, mask=y1, mask_rows=rows)
        assert torch.equal(dx_a[:n], dx_b[:n]), mode


@pytest.mark.parametrize("R,K,J,masked", [(700, 128, 512, True), (1000, 512, 1024, False), (5000, 64, 256, True),
                                          (333, 256, 64, False), (4096, 1024, 512, False), (520, 132, 36, True)])
@pytest.mark.parametrize("mode", ["bf16", "split"])
def test_weight_gradient_on_the_bf16_pipe(mode, R, K, J, masked):
    """catre_op_gemm_tn_bias_lp: dW = (dY .* mask)^T X with bf16 MFMAs.  'bf16' must equal the product of the
    bf16-ROUNDED operands (fp32 re-association only); 'split' (hi + lo, three products) must reach the exact product
    to 2e-5 relative of the largest entry; the bias gradient is an fp32 sum of the unrounded dY in both."""
    from catre_amd import train_ops as T

    g = _gen(R + K + J)
    dy = torch.randn(R, J, generator=g)
    x = torch.randn(R, K, generator=g)
    ym = torch.randn(R, J, generator=g) if masked else None
    dyd, xd = dy.to(DEV), x.to(DEV)
    dw, db = T._gemm_tn(dyd, xd, with_bias=True, ymask=ym.to(DEV) if masked else None, amp=T._MODES[mode])
    dym = dy * (ym > 0) if masked else dy
    rnd = (lambda t: t.to(torch.bfloat16).double()) if mode == "bf16" else (lambda t: t.double())
    want = rnd(dym).t() @ rnd(x)
    scale = float(want.abs().max())
    _cmp(dw, want, f"dW {mode}", atol=2e-5 * scale, rtol=0)
    _cmp(db, dym.double().sum(0), "db", atol=2e-4, rtol=2e-5)
    dw2, _ = T._gemm_tn(dyd, xd, with_bias=True, ymask=ym.to(DEV) if masked else None, amp=T._MODES[mode])
    assert torch.equal(dw, dw2), "deterministic"


@pytest.mark.parametrize("R,J,Kx,Kw,masked", [(512, 512, 1024, 1024, True), (512, 9, 256, 256, False), (256, 3, 1091, 1091, False),
                                              (33, 40, 77, 77, True), (7, 3, 5, 5, True), (100, 64, 8, 3, True),
                                              (65, 36, 3, 8, False), (2047, 256, 64, 64, True), (512, 4096, 256, 256, False)])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_small_linear_backward_in_one_launch(R, J, Kx, Kw, masked, mode):
    """catre_op_fc_bwd: dX, dW and db of a linear layer on < 2048 rows from one launch, any widths and leading dimensions
    (x wider or narrower than the weight: the missing columns count as zeros), ReLU mask on load.  'bf16': dW is the
    product of the bf16-ROUNDED operands, dX and db stay fp32."""
    from catre_amd import hip
    from catre_amd import train_ops as T

    g = _gen(R + J + Kx + Kw)
    ldy, ldx, ldw = J + 3, Kx + 1, Kw + 5           # odd leading dimensions: nothing is aligned
    dyb, ymb = torch.randn(R, ldy, generator=g), torch.randn(R, ldy, generator=g)
    xb, wb = torch.randn(R, ldx, generator=g), torch.randn(J, ldw, generator=g) / Kw ** 0.5
    dy, ym, x, w = dyb[:, :J], ymb[:, :J], xb[:, :Kx], wb[:, :Kw]
    dv = (dy * (ym > 0) if masked else dy).double()
    kk = min(Kx, Kw)
    want_dx = torch.zeros(R, Kx, dtype=torch.float64)
    want_dx[:, :kk] = dv @ w.double()[:, :kk]
    rnd = (lambda t: t.to(torch.bfloat16).double()) if mode == "bf16" else (lambda t: t.double())
    want_dw = torch.zeros(J, Kw, dtype=torch.float64)
    want_dw[:, :kk] = rnd(dv.float()).t() @ rnd(x)[:, :kk]
    dyd, ymd, xd, wd = dyb.to(DEV), ymb.to(DEV), xb.to(DEV), wb.to(DEV)
    dx = torch.full((R, Kx), 7.0, device=DEV)
    dw = torch.full((J, Kw), 7.0, device=DEV)
    db = torch.full((J,), 7.0, device=DEV)
    lib = hip.load()

    def run(dx, dw, db):
        hip.check(lib.catre_op_fc_bwd(hip.ptr(dyd), ldy, hip.ptr(ymd) if masked else None, hip.ptr(xd), ldx, hip.ptr(wd), ldw,
                                      hip.ptr(dx), hip.ptr(dw), hip.ptr(db), R, J, Kx, Kw, T._MODES[mode],
                                      hip.stream_ptr(torch.device(DEV))), "catre_op_fc_bwd")
    run(dx, dw, db)
    _cmp(dx, want_dx, "dx", atol=2e-5 * max(1.0, float(want_dx.abs().max())), rtol=0)
    _cmp(dw, want_dw, "dw", atol=2e-5 * max(1.0, float(want_dw.abs().max())), rtol=0)
    _cmp(db, dv.sum(0), "db", atol=2e-4, rtol=2e-5)
    # each output alone gives the same bits (the roles are separate workgroups), twice gives the same bits
    dx2, dw2, db2 = torch.empty_like(dx), torch.empty_like(dw), torch.empty_like(db)
    run(dx2, None, None)
    run(None, dw2, db2)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)
    assert lib.catre_op_fc_bwd(hip.ptr(dyd), ldy, None, hip.ptr(xd), ldx, hip.ptr(wd), ldw, None, None, hip.ptr(db), R, J, Kx,
                               Kw, 0, hip.stream_ptr(torch.device(DEV))) != 0, "db alone is refused"


def test_hub_sums_the_consumers_of_a_tensor_in_one_launch():
    """train_ops.hub (catre_op_sum_rows): t -> (t[:rc], t, t); the gradient of t is the sum of what its consumers send back,
    equal to autograd's own route through a slice (zero-fill + copy) and two adds - for any subset of consumers that is used,
    and for a slice gradient that arrives as a column slice of a wider tensor (read in place)."""
    from catre_amd import train_ops as T

    g = _gen(11)
    R, rc, K = 37, 13, 70
    base = torch.randn(R, K, generator=g)
    wa, wb = torch.randn(R, K, generator=g).to(DEV), torch.randn(R, K, generator=g).to(DEV)
    wide = torch.randn(rc, K + 9, generator=g).to(DEV)   # the slice consumer's gradient: columns 4 .. 4 + K of a wider matrix
    for use in ((1, 1, 1), (1, 0, 0), (0, 1, 1), (1, 1, 0), (0, 0, 1)):
        t = base.clone().to(DEV).requires_grad_(True)
        c, a, b = T.hub(t, rc)
        assert torch.equal(c, t[:rc]) and torch.equal(a, t) and torch.equal(b, t)
        tr = base.clone().to(DEV).requires_grad_(True)
        loss, lossr = 0.0, 0.0
        if use[0]:
            pad = torch.zeros(rc, K + 9, device=DEV)
            pad = torch.cat([pad[:, :4], c, pad[:, 4 + K:]], 1)       # c's gradient = wide[:, 4:4+K]: a strided view
            loss = loss + (pad * wide).sum()
            lossr = lossr + (tr[:rc] * wide[:, 4:4 + K]).sum()
        if use[1]:
            loss, lossr = loss + (a * wa).sum(), lossr + (tr * wa).sum()
        if use[2]:
            loss, lossr = loss + (b * wb).sum(), lossr + (tr * wb).sum()
        loss.backward()
        lossr.backward()
        _cmp(t.grad, tr.grad.cpu(), f"hub {use}", atol=1e-6, rtol=1e-6)


def test_conv_p_backward_also_gives_the_neck_bias_gradient():
    """catre_op_wsum_bwd_n: dbn = column sums of dY3 taken as (sum_p w_p) (sum_b dout_b) in the launch that sums dbias;
    dY3, dw and dbias are those of catre_op_wsum_bwd bit for bit."""
    from catre_amd import hip

    g = _gen(12)
    B, P = 9, 192
    dout, y3, w = torch.randn(B, 3, generator=g), torch.randn(B * P, 3, generator=g), torch.randn(P, generator=g)
    dd, yd, wd = dout.to(DEV), y3.to(DEV), w.to(DEV)
    lib, st = hip.load(), hip.stream_ptr(torch.device(DEV))
    outs = []
    for with_n in (False, True):
        dy, dw, db, dbn = (torch.empty(B * P, 3, device=DEV), torch.empty(P, device=DEV), torch.empty(1, device=DEV),
                           torch.full((3,), 7.0, device=DEV))
        ws = torch.empty(B * P * 4, dtype=torch.uint8, device=DEV)
        if with_n:
            hip.check(lib.catre_op_wsum_bwd_n(hip.ptr(dd), hip.ptr(yd), hip.ptr(wd), hip.ptr(dy), hip.ptr(dw), hip.ptr(db),
                                              hip.ptr(dbn), 0, hip.ptr(ws), ws.numel(), B, P, st), "catre_op_wsum_bwd_n")
        else:
            hip.check(lib.catre_op_wsum_bwd(hip.ptr(dd), hip.ptr(yd), hip.ptr(wd), hip.ptr(dy), hip.ptr(dw), hip.ptr(db), 0,
                                            hip.ptr(ws), ws.numel(), B, P, st), "catre_op_wsum_bwd")
        outs.append((dy, dw, db, dbn))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    want_dy = (w.double().view(1, P, 1) * dout.double().view(B, 1, 3)).reshape(B * P, 3)
    _cmp(outs[1][0], want_dy, "dY3")
    _cmp(outs[1][3], want_dy.sum(0), "dbn", atol=2e-4, rtol=2e-5)


def test_new_training_entry_points_refuse_bad_arguments():
    """Error behaviour of the round's entry points: a status, not a launch (no fallback, nothing written)."""
    import ctypes

    from catre_amd import hip

    lib, st = hip.load(), hip.stream_ptr(torch.device(DEV))
    a = torch.zeros(64, 64, device=DEV)
    p = hip.ptr
    # fc_bwd: 2048 rows or more belong to the tiled kernels; a leading dimension under the width; no output at all
    assert lib.catre_op_fc_bwd(p(a), 64, None, p(a), 64, p(a), 64, p(a), p(a), None, 2048, 64, 64, 64, 0, st) != 0
    assert lib.catre_op_fc_bwd(p(a), 32, None, p(a), 64, p(a), 64, p(a), p(a), None, 64, 64, 64, 64, 0, st) != 0
    assert lib.catre_op_fc_bwd(p(a), 64, None, p(a), 64, p(a), 64, None, None, None, 64, 64, 64, 64, 0, st) != 0
    assert lib.catre_op_fc_bwd(p(a), 64, None, p(a), 64, p(a), 64, p(a), p(a), None, 64, 64, 64, 64, 7, st) != 0   # dtype
    # sum_rows: the slice cannot be longer than the tensor, its pitch not under the width
    assert lib.catre_op_sum_rows(p(a), None, p(a), 64, p(a), 32, 33, 64, st) != 0
    assert lib.catre_op_sum_rows(p(a), None, p(a), 32, p(a), 32, 16, 64, st) != 0
    # bf16-row flag: only with the bf16-operand kernels
    cnt = torch.ones(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    assert lib.catre_op_gemm_tn_bias_nr(p(a), 64, None, 0, p(a), 64, None, p(a), None, 64, 64, 64, 0, p(ws), ws.numel(), p(cnt),
                                        hip.ROWS_BF16 | 0, st) != 0
    assert lib.catre_op_gemm_rows_nr(p(a), 64, None, 0, p(a), None, p(a), 64, None, p(a), 64, 64, 64, 64, 0, p(cnt),
                                     hip.ROWS_BF16 | 2, st) != 0
    # loss sums: more than six terms, a term index out of range
    terms = (ctypes.c_int32 * 7)(0, 1, 2, 3, 4, 5, 0)
    z = torch.zeros(64, device=DEV)
    args = [p(z)] * 15
    assert lib.catre_loss_fwd_sums(*args, terms, 7, p(z), 2, 4, 1, st) != 0
    bad = (ctypes.c_int32 * 1)(9)
    assert lib.catre_loss_fwd_sums(*args, bad, 1, p(z), 2, 4, 1, st) != 0
    torch.cuda.synchronize()
    assert float(a.abs().max()) == 0.0 and float(z.abs().max()) == 0.0


def test_linear_identity_tail():
    from catre_amd import train_ops as T

    g = _gen(1)
    x, xr = _leaf(torch.randn(6, 256, generator=g))
    w, wr = _leaf(torch.randn(9, 256, generator=g) / 16)
    b, br = _leaf(torch.randn(9, generator=g) * 0.1)
    y = T.linear(x, w, b, identity_k=3)
    yr = F.linear(xr, wr, br) + torch.eye(3, dtype=torch.float64).reshape(1, 9)
    _cmp(y, yr, "y")
    y.sum().backward()
    yr.sum().backward()
    _cmp(x.grad, xr.grad, "dx")
    _cmp(w.grad, wr.grad, "dw")


def _cloud_slices(B, N, M):
    sl = [(b * N, N) for b in range(B)] + [(B * N + b * M, M) for b in range(B)]
    return sl


@pytest.mark.parametrize("dims", [(3, 200, 70, 128, 1024),    # ragged clouds: materialise + pool
                                  (2, 192, 128, 128, 1024),   # 64-aligned clouds: max / arg-max fused into the GEMM
                                  (2, 128, 64, 512, 1024), (3, 64, 0, 64, 256)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("mode", ["fp32", "split"])
def test_linear_maxpool(relu, dims, mode):
    from catre_amd import train_ops as T

    B, N, M, K, J = dims
    g = _gen(7)
    R = B * (N + M)
    x, xr = _leaf(torch.randn(R, K, generator=g))
    w, wr = _leaf(torch.randn(J, K, 1, generator=g) / K ** 0.5)
    b, br = _leaf(torch.randn(J, generator=g) * 0.1)
    with T.amp_mode(mode):
        out = T.linear_maxpool(x, w, b, relu, B, N, M)
    yr = F.linear(xr, wr[:, :, 0], br)
    ref = torch.stack([yr[s:s + n].max(0)[0] for s, n in _cloud_slices(B, N, M) if n > 0])
    ref = ref.relu() if relu else ref
    tol = dict(atol=1e-4, rtol=2e-5) if mode == "split" else {}
    _cmp(out, ref, "pooled", **tol)
    dg = torch.randn(ref.shape[0], J, generator=g)
    out.backward(dg.to(DEV))
    ref.backward(dg.double())
    _cmp(x.grad, xr.grad, "dx", **tol)
    _cmp(w.grad, wr.grad, "dw", atol=1e-4)
    _cmp(b.grad, br.grad, "db", atol=1e-4)


@pytest.mark.parametrize("mode", ["fp32", "bf16", "split"])
@pytest.mark.parametrize("B,N,M", [(3, 128, 64), (2, 192, 0), (3, 100, 60)])
def test_linear_with_per_cloud_bias(mode, B, N, M):
    """rot-head layer 0: x W^T + bias[cloud(row)] with the bias in the GEMM epilogue (ragged clouds: linear + add)."""
    from catre_amd import train_ops as T

    g = _gen(B + N + M)
    P = N + M
    x, xr = _leaf(torch.randn(B * P, 64, generator=g))
    w, wr = _leaf(torch.randn(256, 64, generator=g) / 8)
    nb = 2 * B if M > 0 else B
    bias, br = _leaf(torch.randn(nb, 256, generator=g))
    with T.amp_mode(mode):
        y = T.linear_cloudbias(x, w, bias, B, N, M)
    # (ragged clouds under 2048 rows take linear + rowbias_add on the split-K fp32 linear: no operand rounding there)
    rounded = mode == "bf16" and (T._rot_linear_ok(B * P, 256, 64, N, M) or B * P >= 2048)
    rnd = (lambda t: t.to(torch.bfloat16).double()) if rounded else (lambda t: t)
    bfull = torch.cat([br[:B].unsqueeze(1).expand(B, N, 256)] + ([br[B:].unsqueeze(1).expand(B, M, 256)] if M else []), 1)
    yr = (rnd(xr.float()).double() if rounded else xr) @ (rnd(wr.float()).double() if rounded else wr).t()
    yr = yr + bfull.reshape(B * P, 256)
    tol = dict(atol=1e-4, rtol=2e-5) if mode != "fp32" else {}
    _cmp(y, yr, "y", **tol)
    if mode == "bf16":
        return
    dy = torch.randn(B * P, 256, generator=g)
    y.backward(dy.to(DEV))
    yr.backward(dy.double())
    _cmp(x.grad, xr.grad, "dx", **tol)
    _cmp(w.grad, wr.grad, "dw", atol=2e-3 if mode == "split" else 2e-4, rtol=2e-5)
    _cmp(bias.grad, br.grad, "dbias", atol=2e-4, rtol=2e-5)


@pytest.mark.parametrize("mode", ["fp32", "split"])
def test_groupnorm_statistics_from_the_gemm_epilogue(mode):
    """linear (+ per-cloud bias) emits per-tile GroupNorm partials; gn_points_gelu fed with them must equal the version
    that makes its own statistics pass, forward and backward."""
    from catre_amd import train_ops as T

    g = _gen(77)
    B, N, M = 3, 128, 64
    P = N + M
    x = torch.randn(B * P, 64, generator=g).to(DEV)
    w0 = (torch.randn(256, 64, generator=g) / 8).to(DEV).requires_grad_(True)
    w1 = (torch.randn(256, 256, generator=g) / 16).to(DEV).requires_grad_(True)
    b1 = (torch.randn(256, generator=g) * 0.1 + 3.0).to(DEV).requires_grad_(True)  # a large mean: the shift matters
    bias = torch.randn(2 * B, 256, generator=g).to(DEV)
    ga = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV).requires_grad_(True)
    be = (0.1 * torch.randn(256, generator=g)).to(DEV).requires_grad_(True)
    dout = torch.randn(B * P, 256, generator=g).to(DEV)

    def run(fused):
        for t in (w0, w1, b1, ga, be):
            t.grad = None
        with T.amp_mode(mode):
            if fused:
                y, part = T.linear_cloudbias(x, w0, bias, B, N, M, with_gn_partials=True)
                assert part is not None and part.shape == (B * P // 64, 32, 2)
                a = T.gn_points_gelu(y, ga, be, B, P, part)
                y, part = T.linear_gn_partials(a, w1, b1, B, N, M)
                assert part is not None
                a = T.gn_points_gelu(y, ga, be, B, P, part)
            else:
                y = T.rowbias_add(T.linear(x, w0, None), bias, B, N, M)
                a = T.gn_points_gelu(y, ga, be, B, P)
                a = T.gn_points_gelu(T.linear(a, w1, b1), ga, be, B, P)
        (a * dout).sum().backward()
        return a.detach().clone(), [t.grad.clone() for t in (w0, w1, b1, ga, be)]

    a1, g1 = run(True)
    a2, g2 = run(False)
    # same statistics up to the order of the partial sums (split: the unfused chain's 576-row linears run on the fp32
    # split-K kernel, the fused one on split-bf16 products)
    assert (a1 - a2).abs().max() < (1e-4 if mode == "split" else 2e-5)
    for u, v in zip(g1, g2):
        assert (u - v).abs().max() <= (1e-4 if mode == "split" else 2e-5) * float(v.abs().max()) + 1e-7


def test_maxpool_points_and_cloud_matmul():
    from catre_amd import train_ops as T

    B, N, M = 2, 130, 45
    R = B * (N + M)
    g = _gen(9)
    for kd, oc in ((3, 8), (64, 64)):
        x, xr = _leaf(torch.randn(R, kd, generator=g))
        Tm, Tr = _leaf(torch.randn(2 * B, kd, kd, generator=g) / kd ** 0.5 + torch.eye(kd))
        y = T.cloud_matmul(x, Tm, B, N, M, out_cols=oc)
        ref = torch.cat([xr[s:s + n] @ Tr[c] for c, (s, n) in enumerate(_cloud_slices(B, N, M))])
        _cmp(y[:, :kd], ref, f"cloud_matmul {kd}")
        if oc > kd:
            assert float(y[:, kd:].abs().max()) == 0.0
        p = T.maxpool_points(y, B, N, M)
        pref = torch.stack([ref[s:s + n].max(0)[0] for s, n in _cloud_slices(B, N, M)])
        _cmp(p[:, :kd], pref, "maxpool")
        dy = torch.randn(R, oc, generator=g)
        dp = torch.randn(2 * B, oc, generator=g)
        (y * dy.to(DEV)).sum().backward(retain_graph=True)
        (p * dp.to(DEV)).sum().backward()
        ((ref * dy[:, :kd].double()).sum() + (pref * dp[:, :kd].double()).sum()).backward()
        _cmp(x.grad, xr.grad, f"dx {kd}")
        _cmp(Tm.grad, Tr.grad, f"dT {kd}", atol=1e-4)


def test_rowbias_gn_points_wsum():
    from catre_amd import train_ops as T

    B, N, M = 3, 100, 60
    P = N + M
    g = _gen(11)
    y, yr = _leaf(torch.randn(B * P, 256, generator=g))
    bias, biasr = _leaf(torch.randn(2 * B, 256, generator=g))
    ga, gar = _leaf(1 + 0.1 * torch.randn(256, generator=g))
    be, ber = _leaf(0.1 * torch.randn(256, generator=g))
    wn, wnr = _leaf(torch.randn(3, 256, 1, generator=g) / 16)
    bn, bnr = _leaf(torch.randn(3, generator=g) * 0.1)
    wp, wpr = _leaf(torch.rand(1, P, 1, generator=g) / P)
    bp, bpr = _leaf(torch.randn(1, generator=g) * 0.1)
    z = T.rowbias_add(y, bias, B, N, M)
    a = T.gn_points_gelu(z, ga, be, B, P)
    y3 = T.linear(a, wn, bn)
    out = T.weighted_point_sum(y3, wp, bp, B, P)
    # reference: [B,256,P] layout like the reference rot head
    zr = yr.reshape(B, P, 256)
    bfull = torch.cat([biasr[:B].unsqueeze(1).expand(B, N, 256), biasr[B:].unsqueeze(1).expand(B, M, 256)], 1)
    zr = (zr + bfull).permute(0, 2, 1)
    ar = F.gelu(F.group_norm(zr, 32, gar, ber, 1e-5))
    y3r = F.conv1d(ar, wnr, bnr)                         # [B,3,P]
    outr = F.conv1d(y3r.permute(0, 2, 1), wpr, bpr).squeeze(1)   # [B,3]
    _cmp(out, outr, "out", atol=1e-5)
    d = torch.randn(B, 3, generator=g)
    out.backward(d.to(DEV))
    outr.backward(d.double())
    for got, want, nm in ((y, yr, "dy"), (bias, biasr, "dbias"), (ga, gar, "dgamma"), (be, ber, "dbeta"),
                          (wn, wnr, "dneck"), (bn, bnr, "dneck_b"), (wp, wpr, "dwp"), (bp, bpr, "dbp")):
        _cmp(got.grad, want.grad, nm, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("rd", [3, 2])
def test_gn_gelu_neck_fused_op_matches_fp64_reference(rd):
    """GroupNorm + GELU + neck conv as one op (no [R,256] activation in between): forward and every gradient against
    fp64 torch, on the rot head's own composition linear -> GN -> GELU -> neck -> conv_p (conv_out_per_rot_head.py:126-140)."""
    from catre_amd import train_ops as T
    from catre_amd.heads import neck_weight3

    B, N, M = 3, 128, 64
    P = N + M
    g = _gen(21 + rd)
    a0, a0r = _leaf(torch.randn(B * P, 256, generator=g))
    w1, w1r = _leaf(torch.randn(256, 256, generator=g) / 16)
    b1, b1r = _leaf(torch.randn(256, generator=g) * 0.1 + 2.0)
    ga, gar = _leaf(1 + 0.1 * torch.randn(256, generator=g))
    be, ber = _leaf(0.1 * torch.randn(256, generator=g))
    wn, wnr = _leaf(torch.randn(rd, 256, 1, generator=g) / 16)
    bn, bnr = _leaf(torch.randn(rd, generator=g) * 0.1)
    wp, wpr = _leaf(torch.rand(1, P, 1, generator=g) / P)
    y, part = T.linear_gn_partials(a0, w1, b1, B, N, M)
    assert part is not None
    w3, b3 = neck_weight3(wn, bn)
    y3 = T.gn_points_gelu_neck(y, ga, be, w3, b3, B, P, part)
    assert y3.shape == (B * P, 3)
    out = T.weighted_point_sum(y3, wp, None, B, P)[:, :rd]
    yr = (a0r @ w1r.t() + b1r).reshape(B, P, 256).permute(0, 2, 1)
    ar = F.gelu(F.group_norm(yr, 32, gar, ber, 1e-5))
    y3r = F.conv1d(ar, wnr, bnr)                                  # [B,rd,P]
    outr = F.conv1d(y3r.permute(0, 2, 1), wpr).squeeze(1)         # [B,rd]
    _cmp(y3[:, :rd], y3r.permute(0, 2, 1).reshape(B * P, rd), "y3", atol=1e-5)
    if rd < 3:
        assert float(y3[:, rd:].abs().max()) == 0.0
    _cmp(out, outr, "out", atol=1e-5)
    d = torch.randn(B, rd, generator=g)
    out.backward(d.to(DEV))
    outr.backward(d.double())
    for got, want, nm in ((a0, a0r, "da0"), (w1, w1r, "dw1"), (b1, b1r, "db1"), (ga, gar, "dgamma"), (be, ber, "dbeta"),
                          (wn, wnr, "dneck"), (bn, bnr, "dneck_b"), (wp, wpr, "dwp")):
        _cmp(got.grad, want.grad, nm, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("B,N,M", [(3, 128, 64), (2, 64, 0), (5, 192, 320)])
def test_rot_l0_block_fused_backward_matches_fp64_reference(B, N, M):
    """Layer-0 block of a RotHead (per-cloud-bias linear -> GroupNorm -> GELU) with the one-pass backward: output and
    every gradient against fp64 torch, and against the unfused op chain."""
    from catre_amd import train_ops as T

    P = N + M
    g = _gen(100 + B)
    x, xr = _leaf(torch.randn(B * P, 64, generator=g))
    w, wr = _leaf(torch.randn(256, 64, generator=g) / 8)
    bias, biasr = _leaf(torch.randn(2 * B if M else B, 256, generator=g) * 0.5 + 1.0)
    ga, gar = _leaf(1 + 0.1 * torch.randn(256, generator=g))
    be, ber = _leaf(0.1 * torch.randn(256, generator=g))
    dout = torch.randn(B * P, 256, generator=g)
    assert T.rot_l0_block_ok(x, w, N, M)
    a = T.rot_l0_block(x, w, bias, ga, be, B, N, M)
    yr = (xr @ wr.t()).reshape(B, P, 256)
    bfull = biasr[:B].unsqueeze(1).expand(B, N, 256)
    if M:
        bfull = torch.cat([bfull, biasr[B:].unsqueeze(1).expand(B, M, 256)], 1)
    ar = F.gelu(F.group_norm((yr + bfull).permute(0, 2, 1), 32, gar, ber, 1e-5)).permute(0, 2, 1).reshape(B * P, 256)
    _cmp(a, ar, "a", atol=2e-5)
    a.backward(dout.to(DEV))
    ar.backward(dout.double())
    fused = [t.grad.clone() for t in (x, w, bias, ga, be)]
    for got, want, nm in zip(fused, (xr, wr, biasr, gar, ber), ("dx", "dw", "dbias", "dgamma", "dbeta")):
        _cmp(got, want.grad, nm, atol=1e-4, rtol=1e-4)
    for t in (x, w, bias, ga, be):
        t.grad = None
    y, part = T.linear_cloudbias(x, w, bias, B, N, M, with_gn_partials=True)
    a2 = T.gn_points_gelu(y, ga, be, B, P, part)
    assert (a2 - a).abs().max() <= 1e-5  # (small R takes the untiled linear + its own statistics pass: last-bit differences)
    a2.backward(dout.to(DEV))
    for u, t, nm in zip(fused, (x, w, bias, ga, be), ("dx", "dw", "dbias", "dgamma", "dbeta")):
        assert (u - t.grad).abs().max() <= 2e-5 * float(t.grad.abs().max()) + 1e-6, nm


@pytest.mark.parametrize("B,N,M,rd", [(3, 128, 64, 3), (2, 64, 0, 2), (5, 192, 320, 3), (300, 64, 64, 3)])
def test_rot_l1_block_fused_backward_matches_fp64_reference(B, N, M, rd):
    """Second block of a RotHead (linear -> GroupNorm -> GELU -> neck) with the one-pass backward: output and every
    gradient against fp64 torch, and against the unfused op chain.  (B = 300: more workgroups than CUs, one tile chunk
    per object; B = 3: several chunks per object.)"""
    from catre_amd import train_ops as T
    from catre_amd.heads import neck_weight3

    P = N + M
    g = _gen(200 + B)
    a, ar = _leaf(torch.randn(B * P, 256, generator=g))
    w, wr = _leaf(torch.randn(256, 256, generator=g) / 16)
    b, br = _leaf(torch.randn(256, generator=g) * 0.1 + 1.0)
    ga, gar = _leaf(1 + 0.1 * torch.randn(256, generator=g))
    be, ber = _leaf(0.1 * torch.randn(256, generator=g))
    wn, wnr = _leaf(torch.randn(rd, 256, 1, generator=g) / 16)
    bn, bnr = _leaf(torch.randn(rd, generator=g) * 0.1)
    dout = torch.randn(B * P, 3, generator=g)
    dout[:, rd:] = 0
    assert T.rot_l1_block_ok(a, w, N, M)
    w3, b3 = neck_weight3(wn, bn)
    y3 = T.rot_l1_block(a, w, b, ga, be, w3, b3, B, N, M)
    yr = (ar @ wr.t() + br).reshape(B, P, 256).permute(0, 2, 1)
    y3r = F.conv1d(F.gelu(F.group_norm(yr, 32, gar, ber, 1e-5)), wnr, bnr).permute(0, 2, 1).reshape(B * P, rd)
    _cmp(y3[:, :rd], y3r, "y3", atol=2e-5)
    y3.backward(dout.to(DEV))
    y3r.backward(dout[:, :rd].double())
    leaves, refs = (a, w, b, ga, be, wn, bn), (ar, wr, br, gar, ber, wnr, bnr)
    names = ("da", "dw", "db", "dgamma", "dbeta", "dneck", "dneck_b")
    fused = [t.grad.clone() for t in leaves]
    for got, want, nm in zip(fused, refs, names):
        _cmp(got, want.grad, nm, atol=2e-4 if B > 100 else 1e-4, rtol=1e-4)
    for t in leaves:
        t.grad = None
    y, part = T.linear_gn_partials(a, w, b, B, N, M)
    w3, b3 = neck_weight3(wn, bn)
    y3u = T.gn_points_gelu_neck(y, ga, be, w3, b3, B, P, part)
    assert (y3u - y3).abs().max() <= 1e-6
    y3u.backward(dout.to(DEV))
    for u, t, nm in zip(fused, leaves, names):
        assert (u - t.grad).abs().max() <= 2e-5 * float(t.grad.abs().max()) + 1e-6, nm


def test_gn_rows_gelu():
    from catre_amd import train_ops as T

    g = _gen(12)
    y, yr = _leaf(torch.randn(7, 256, generator=g))
    ga, gar = _leaf(1 + 0.1 * torch.randn(256, generator=g))
    be, ber = _leaf(0.1 * torch.randn(256, generator=g))
    a = T.gn_rows_gelu(y, ga, be)
    ar = F.gelu(F.group_norm(yr, 32, gar, ber, 1e-5))
    _cmp(a, ar, "a")
    d = torch.randn(7, 256, generator=g)
    a.backward(d.to(DEV))
    ar.backward(d.double())
    _cmp(y.grad, yr.grad, "dy", rtol=1e-4)
    _cmp(ga.grad, gar.grad, "dgamma", rtol=1e-4)
    _cmp(be.grad, ber.grad, "dbeta", rtol=1e-4)


@pytest.mark.parametrize("space,zstyle,ka,stype", [("image", "cosypose", True, "iter_add"), ("image", "deepim", False, "mean_mul"),
                                                   ("3D", "cosypose", True, "iter_add")])
@pytest.mark.parametrize("allo", [False, True])
def test_pose_update_bwd(space, zstyle, ka, stype, allo):
    from catre_amd import hip
    from catre_amd import train_ops as T
    from oracle import catre_oracle as O

    g = _gen(13)
    B = 9
    r6, r6r = _leaf(torch.randn(B, 6, generator=g))
    dt, dtr = _leaf(torch.tensor([0.0, 0.0, 1.0]) + 0.05 * torch.randn(B, 3, generator=g))
    ds, dsr = _leaf(0.05 * torch.randn(B, 3, generator=g))
    R0 = O.quat2mat_torch(torch.randn(B, 4, generator=g))
    t0 = torch.tensor([0.0, 0.0, 1.0]) + 0.2 * torch.randn(B, 3, generator=g)
    pose0 = torch.cat([R0, t0.unsqueeze(-1)], -1)
    s0 = 0.1 + 0.1 * torch.rand(B, 3, generator=g)
    ms = 0.1 + 0.1 * torch.rand(B, 3, generator=g)
    K = torch.tensor([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]]).repeat(B, 1, 1)
    o = hip.CatreOpts()
    o.delta_t_space_3d = int(space == "3D"); o.delta_z_deepim = int(zstyle != "cosypose"); o.k_aware = int(ka)
    o.scale_mul = int("add" not in stype); o.scale_base_mean = int("iter" not in stype); o.refine_scale = 1
    o.delta_t_weight = 0.7; o.allo_eps = 1e-4; o.is_allo = int(allo)
    pose, scale = T.pose_update_autograd(r6, dt, ds, pose0.to(DEV), s0.to(DEV), ms.to(DEV), K.to(DEV), o)
    Rr, tr, sr = O.pose_scale_from_delta_init(
        O.rot6d_to_mat_batch(r6r), dtr, dsr, R0.double(), t0.double(), (s0 if "iter" in stype else ms).double(),
        Ks=K.double(), K_aware=ka, delta_T_space=space, delta_T_weight=0.7, delta_z_style=zstyle, scale_type=stype,
        is_allo=allo)
    _cmp(pose[:, :, :3], Rr, "R", atol=3e-6)
    _cmp(pose[:, :, 3], tr, "t", atol=3e-6)
    gp = torch.randn(B, 3, 4, generator=g)
    gs = torch.randn(B, 3, generator=g)
    ((pose * gp.to(DEV)).sum() + (scale * gs.to(DEV)).sum()).backward()
    ((Rr * gp[:, :, :3].double()).sum() + (tr * gp[:, :, 3].double()).sum() + (sr * gs.double()).sum()).backward()
    _cmp(r6.grad, r6r.grad, "d_rot6d", atol=1e-4, rtol=1e-4)
    _cmp(dt.grad, dtr.grad, "d_dt", atol=1e-4, rtol=1e-4)
    _cmp(ds.grad, dsr.grad, "d_ds", atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("B,N,M,K0,J1,J2,J3,relu_pool", [
    (3, 128, 64, 64, 128, 512, 1024, False),    # the trunk tail: pointfeat -> conv2 -> conv3 -> conv4 + max
    (4, 192, 128, 64, 64, 128, 1024, True),     # fstn: h1 -> conv1 -> conv2 -> conv3 + max + ReLU
    (3, 64, 256, 3, 64, 128, 1024, True),       # stn: points (3 columns, no input gradient) -> ...
    (40, 64, 64, 64, 128, 512, 1024, False),    # 5120 rows: several row tiles / wgrad splits past the live-row count
])
@pytest.mark.parametrize("mode", ["fp32", "bf16", "split"])
def test_pooled_chain_row_sparse_backward_matches_fp64_reference(B, N, M, K0, J1, J2, J3, relu_pool, mode):
    """`pooled_chain` (one node for conv -> conv -> conv + max-pool with the ROW-SPARSE backward: live rows compacted on the
    device, dgrad / wgrad on those rows only) vs fp64 torch autograd of the same three layers, and vs the layer-wise HIP ops
    it replaces.  The forward outputs come from the layer-wise ops (what the fused kernels save).  Modes bf16 / split: the
    same GEMM kernels with a device-side row count on the reduced-precision pipes, against fp64 at the mode's own precision."""
    from catre_amd import train_ops as T

    tol64 = {"fp32": 2e-4, "split": 1e-3, "bf16": 4e-2}[mode]

    g = _gen(B * 7 + N + J2)
    R = B * (N + M)
    x, xr = _leaf(torch.randn(R, K0, generator=g))
    if K0 == 3:
        x = x.detach()   # the STN's input points carry no gradient
    ws = []
    for (j, k) in ((J1, K0), (J2, J1), (J3, J2)):
        ws.append(_leaf(torch.randn(j, k, 1, generator=g) / k ** 0.5))
        ws.append(_leaf(0.1 * torch.randn(j, generator=g)))
    (w1, w1r), (b1, b1r), (w2, w2r), (b2, b2r), (w3, w3r), (b3, b3r) = ws
    Gd = torch.randn(2 * B, J3, generator=g)

    def cloud_max(y):
        parts = [y[: B * N].view(B, N, -1).max(1)[0], y[B * N:].view(B, M, -1).max(1)[0]]
        return torch.cat(parts, 0)

    # fp64 reference
    y1r = F.relu(F.linear(xr, w1r[:, :, 0], b1r))
    y2r = F.relu(F.linear(y1r, w2r[:, :, 0], b2r))
    gr = cloud_max(F.linear(y2r, w3r[:, :, 0], b3r))
    if relu_pool:
        gr = F.relu(gr)
    (gr * Gd.double()).sum().backward()

    # layer-wise HIP ops: forward values for `pre`, and their own gradients for comparison
    y1 = T.linear(x, w1, b1, relu=True)
    y2 = T.linear(y1, w2, b2, relu=True)
    gl = T.linear_maxpool(y2, w3, b3, relu_pool, B, N, M)
    (gl * Gd.to(DEV)).sum().backward()
    lw = {k: v.grad.clone() for k, v in (("w1", w1), ("b1", b1), ("w2", w2), ("b2", b2), ("w3", w3), ("b3", b3))}
    lx = x.grad.clone() if x.requires_grad else None
    for t in (w1, b1, w2, b2, w3, b3) + ((x,) if x.requires_grad else ()):
        t.grad = None
    # (g without the pooled ReLU, idx) the way the fused kernels hand them over.  The arg-max rows are the fp64 reference's:
    # in the reduced-precision modes a rounded product can flip single decisions, and this test is about the backward
    with torch.no_grad():
        ypre_r = F.linear(y2r.detach(), w3r.detach()[:, :, 0], b3r.detach())
        idx_o = ypre_r[: B * N].view(B, N, -1).argmax(1).int() + (torch.arange(B).int() * N)[:, None]
        idx_p = ypre_r[B * N:].view(B, M, -1).argmax(1).int() + (B * N + torch.arange(B).int() * M)[:, None]
        idx = torch.cat([idx_o, idx_p], 0).contiguous().to(DEV)
        ypre = F.linear(y2.detach().double(), w3.detach().double()[:, :, 0], b3.detach().double())
        gp = ypre.gather(0, idx.long()).float().contiguous()
    # the saved activations are the fp32 ops' in every mode (a rounded forward would flip ReLU masks next to zero, which is
    # not what this test is about); the chain's GEMMs run on the pipe of `mode`
    ctx_mode = T.amp_mode(mode)
    ctx_mode.__enter__()
    assert T.pooled_chain_ok(x, w1, w2, w3, N, M)
    gc = T.pooled_chain(x, w1, b1, w2, b2, w3, b3, relu_pool, B, N, M, (y1.detach(), y2.detach(), gp, idx))
    _cmp(gc, gr, "pooled output", atol=1e-4 if mode == "fp32" else 5e-2, rtol=1e-4 if mode == "fp32" else 5e-2)
    (gc * Gd.to(DEV)).sum().backward()
    ctx_mode.__exit__(None, None, None)
    for name, t, tr in (("w1", w1, w1r), ("b1", b1, b1r), ("w2", w2, w2r), ("b2", b2, b2r), ("w3", w3, w3r), ("b3", b3, b3r)):
        scale = float(tr.grad.abs().max()) + 1e-12
        err = float((t.grad.cpu().double() - tr.grad).abs().max()) / scale
        assert err <= tol64, (name, err)
        if mode == "fp32":
            errl = float((t.grad - lw[name]).abs().max()) / scale
            assert errl <= 2e-4, (name, "vs layer-wise", errl)
    if x.requires_grad:
        scale = float(xr.grad.abs().max()) + 1e-12
        assert float((x.grad.cpu().double() - xr.grad).abs().max()) / scale <= tol64
        # rows that are nobody's arg-max get EXACT zeros, the others the layer-wise values up to re-association
        live = torch.zeros(R, dtype=torch.bool, device=DEV)
        live[idx.long().reshape(-1)] = True
        assert float(x.grad[~live].abs().max()) == 0.0
        if mode == "fp32":
            assert float((x.grad - lx).abs().max()) / scale <= 2e-4
    else:
        assert x.grad is None


def test_pooled_chain_with_no_live_row_returns_exact_zeros():
    """Upstream gradient identically zero: no row is live, the device-side count is 0, every tile / split of the row GEMMs
    leaves at once - all gradients are exact zeros (nothing uninitialised leaks out of the worst-case-sized buffers)."""
    from catre_amd import train_ops as T

    B, N, M, K0, J1, J2, J3 = 3, 128, 64, 64, 128, 512, 1024
    g = _gen(5)
    R = B * (N + M)
    x = torch.randn(R, K0, generator=g).to(DEV).requires_grad_(True)
    ws = [torch.randn(j, k, 1, generator=g).div(k ** 0.5).to(DEV).requires_grad_(True) for (j, k) in ((J1, K0), (J2, J1), (J3, J2))]
    bs = [(0.1 * torch.randn(j, generator=g)).to(DEV).requires_grad_(True) for j in (J1, J2, J3)]
    with torch.no_grad():
        y1 = T.linear(x, ws[0], bs[0], relu=True)
        y2 = T.linear(y1, ws[1], bs[1], relu=True)
        gp = T.linear_maxpool(y2, ws[2], bs[2], False, B, N, M)
    idx = torch.zeros(2 * B, J3, dtype=torch.int32, device=DEV)
    idx[:B] = (torch.arange(B, device=DEV).int() * N)[:, None]
    idx[B:] = (B * N + torch.arange(B, device=DEV).int() * M)[:, None]
    out = T.pooled_chain(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], False, B, N, M, (y1, y2, gp, idx))
    (out * 0.0).sum().backward()
    for t in [x] + ws + bs:
        assert t.grad is not None and float(t.grad.abs().max()) == 0.0 and torch.isfinite(t.grad).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp32", "split"])
@pytest.mark.parametrize("B,N,M", [(3, 128, 64), (2, 64, 192), (5, 256, 256), (1, 64, 64), (2, 2048, 1024)])
def test_fused_rot_heads_forward_matches_the_per_head_blocks(B, N, M, mode):
    """Both RotHeads' forward on the fused inference kernels with saves (`catre_train_rot_fwd`: GroupNorm-0 statistics from
    pointfeat moments, layer 0 + GN0 + GELU + layer 1 per tile) against the per-head path (`_rot_head`: row GEMM ->
    GroupNorm/GELU pass -> row GEMM -> neck), heads/conv_out_per_rot_head.py:126-140: the 6-d rotation output and every
    gradient (all head parameters, the global feature, pointfeat).  fp32: one graph node (train_ops._RotHeads, GroupNorm-1
    sums from forward-side moments); split: the per-head nodes around the fused kernel's buffers (`pre=`)."""
    from catre_amd import synth, train_forward as TF
    from catre_amd import train_ops as T
    from catre_amd.CATRE_disR_shared import build_model_optimizer, expected_state_shapes
    from catre_amd.config import default_cfg

    cfg = default_cfg(num_pcl=N, num_kps=M, device="cuda:0")
    model, _ = build_model_optimizer(cfg, is_test=False)
    model.load_state_dict({k: v.cuda() for k, v in synth.recipe_state_dict(expected_state_shapes(cfg)).items()})
    model.train()
    rt = model._runtime()
    p = dict(model.named_parameters())
    gen = torch.Generator().manual_seed(B * 1000 + N + M)
    R = B * (N + M)
    pf0 = (torch.randn(R, 64, generator=gen) * 0.7).cuda()
    g0 = torch.randn(2 * B, 1024, generator=gen).cuda().relu()
    w6 = torch.randn(B, 6, generator=gen).cuda()

    def run(fused):
        for q in p.values():
            q.grad = None
        pf = pf0.clone().requires_grad_(True)
        g = g0.clone().requires_grad_(True)
        with T.amp_mode(mode):
            pf_obj = T.object_major(pf, B, N, M)
            if fused:
                rt._fingerprint = None   # what the first encoder kernel of a training forward does: re-pack the weight image
                fn = TF._rot_heads_fused if mode == "fp32" else TF._rot_heads_split
                rx, ry = fn(g, pf, pf_obj, p, rt, B, N, M)
            else:
                rx = TF._rot_head(g, pf_obj, p, "rot_head.rot_head_x", B, N, M)
                ry = TF._rot_head(g, pf_obj, p, "rot_head.rot_head_y", B, N, M)
            out = torch.cat([rx, ry], 1)
            (out * w6).sum().backward()
        grads = {k: q.grad.clone() for k, q in p.items() if q.grad is not None}
        grads["pointfeat"], grads["g"] = pf.grad.clone(), g.grad.clone()
        return out.detach(), grads

    with T.amp_mode(mode):
        assert TF._rot_heads_fused_ok(p, pf0, N, M) == (mode == "fp32") and TF._rot_heads_shapes_ok(p, pf0, N, M)
    want, gw = run(False)
    got, gg = run(True)
    assert torch.isfinite(got).all()
    # forward: same arithmetic except GN0 statistics (moment form vs tile partials): 1e-5 relative to the output scale
    assert (got - want).abs().max() <= 2e-5 * want.abs().max().clamp_min(1.0), (got - want).abs().max()
    assert set(gg) == set(gw) and len(gg) >= 2 * 10 + 2, sorted(gg)
    for k in gw:
        err = float((gg[k] - gw[k]).norm() / gw[k].norm().clamp_min(1e-12))
        assert err <= 2e-4, (k, err)
