"""The autocast rotation heads' fused backward on the bf16 matrix pipe (k_rot_l1_bwd_bf, train_ops._RotL1TailLP) against the
layer-wise autocast ops it replaces (same operand roundings: only the fp32 summation order differs) and against the fp32
fused op (bf16-operand tolerance).  Reference behaviour: heads/conv_out_per_rot_head.py:129-140 under engine.py:304's
autocast."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _head_tensors(B, N, M, seed):
    g = torch.Generator().manual_seed(seed)
    P = N + M
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    t = dict(a=r(B * P, 256), w=r(256, 256, 1, sc=0.06), b=r(256, sc=0.1), gamma=1 + r(256, sc=0.1), beta=r(256, sc=0.1),
             wn=r(3, 256, sc=0.1), bn=r(3, sc=0.1), wp=r(1, P, 1, sc=0.05), bp=r(1, sc=0.1), dout=r(B, 3))
    for k in t:
        if k != "dout":
            t[k].requires_grad_(True)
    return t


def _run(fn, t):
    for v in t.values():
        v.grad = None
    out = fn()
    out.backward(t["dout"])
    return out.detach().clone(), {k: v.grad.detach().clone() for k, v in t.items() if k != "dout"}


@pytest.mark.parametrize("B,N,M", [(3, 128, 64), (5, 256, 192), (2, 1024, 1024)])
def test_fused_lp_tail_matches_the_layerwise_autocast_ops(B, N, M):
    from catre_amd import train_ops as T

    t = _head_tensors(B, N, M, 100 + B)
    P = N + M

    def fused():
        return T.rot_l1_tail_lp(t["a"], t["w"], t["b"], t["gamma"], t["beta"], t["wn"], t["bn"], t["wp"], t["bp"], B, N, M)

    def layerwise():
        y, part = T.linear_gn_partials(t["a"], t["w"], t["b"], B, N, M)
        return T.neck_tail(y, t["gamma"], t["beta"], t["wn"], t["bn"], t["wp"], t["bp"], B, P, part)

    with T.amp_mode("bf16"):
        assert T.rot_l1_tail_lp_ok(t["a"], t["w"], t["b"], N, M)
        of, gf = _run(fused, t)
        ol, gl = _run(layerwise, t)
    with T.amp_mode("fp32"):
        o32, g32 = _run(layerwise, t)
    assert torch.equal(of, ol), "the fused node's forward is the layer-wise autocast forward"
    for k in gl:
        scale = float(gl[k].abs().max()) + 1e-30
        err = float((gf[k] - gl[k]).abs().max()) / scale
        # same bf16 operands, fp32 accumulation in a different order (under 2048 rows the layer-wise data gradient is the
        # fp32 split-K linear, not a bf16-operand GEMM: bf16-operand tolerance there)
        tol = (5e-3 if B * P < 2048 else 2e-4) if k == "a" else (2e-4 if k in ("w", "b") else 1e-6)
        assert err <= tol, (k, err)
        err32 = float((gf[k] - g32[k]).abs().max()) / (float(g32[k].abs().max()) + 1e-30)
        assert err32 <= 3e-2, (k, err32)


def test_autocast_iteration_with_fused_lp_heads_tracks_the_layerwise_iteration():
    """Whole training iteration under torch.autocast with and without the fused nodes: same losses (forward is the same
    kernels), every gradient within 2e-3 of its largest entry."""
    from catre_amd import train_ops as T
    from test_hip_train import _train_setup, _iteration

    B, N, M = 6, 256, 192
    cfg, sd, kw, sym, ((model, opt),) = _train_setup(B, N, M, 31, 1)
    res = []
    for fused in (True, False):
        # (per-model knobs: cfg.MODEL.CATRE.TRAIN_KERNELS) the fp32-row form: the layer-wise path's own tensors
        model.cfg.MODEL.CATRE.TRAIN_KERNELS = dict(fused_lp_rot=fused, lp_rot_bf16_rows=False)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ld = _iteration(model, kw, sym)
        res.append((ld, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    (lf, gf), (ll, gl) = res
    for k in ll:
        assert torch.equal(lf[k], ll[k]), k
    assert gf.keys() == gl.keys()
    for k in gl:
        scale = float(gl[k].abs().max()) + 1e-30
        assert float((gf[k] - gl[k]).abs().max()) / scale <= 2e-3, k


@pytest.mark.parametrize("B,N,M", [(5, 256, 192), (2, 1024, 1024)])
def test_l0_block_under_autocast_matches_the_layerwise_autocast_ops(B, N, M):
    """train_ops._RotL0Block under autocast (bf16-operand forward linear, one-pass backward) against linear_cloudbias +
    gn_points_gelu in the same mode: identical forward, gradients within the bf16-operand tolerance of each other and of the
    fp32 block."""
    from catre_amd import train_ops as T

    g = torch.Generator().manual_seed(7 + B)
    P = N + M
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    t = dict(x=r(B * P, 64), w=r(256, 64, 1, sc=0.1), bias=r(2 * B, 256, sc=0.3), gamma=1 + r(256, sc=0.1), beta=r(256, sc=0.1))
    for v in t.values():
        v.requires_grad_(True)
    t["dout"] = r(B * P, 256)

    def fused():
        return T.rot_l0_block(t["x"], t["w"], t["bias"], t["gamma"], t["beta"], B, N, M)

    def layerwise():
        y, part = T.linear_cloudbias(t["x"], t["w"], t["bias"], B, N, M, with_gn_partials=True)
        return T.gn_points_gelu(y, t["gamma"], t["beta"], B, P, part)

    with T.amp_mode("bf16"):
        assert T.rot_l0_block_ok(t["x"], t["w"], N, M)
        of, gf = _run(fused, t)
        ol, gl = _run(layerwise, t)
    with T.amp_mode("fp32"):
        o32, g32 = _run(fused, t)
    assert torch.equal(of, ol)
    for k in gl:
        scale = float(g32[k].abs().max()) + 1e-30
        # same bf16 operand roundings as the layer-wise GEMMs: only the fp32 summation order differs
        assert float((gf[k] - gl[k]).abs().max()) / scale <= 5e-4, (k, float((gf[k] - gl[k]).abs().max()) / scale)
        assert float((gf[k] - g32[k]).abs().max()) / scale <= 3e-2, k


@pytest.mark.parametrize("B,N,M", [(5, 256, 192), (2, 1024, 1024), (40, 1024, 1024)])
def test_head_with_bf16_rows_tracks_the_fp32_row_form(B, N, M):
    """train_ops._RotHeadLP (y0 / a0 / y1 / dA as bf16 rows) against the same head with fp32 rows (_RotL0Block + _RotL1TailLP,
    same bf16-operand GEMMs) and against the fp32 head: a0 is rounded to bf16 by the next GEMM in both forms, so only the
    rounding of y0, y1 and dA is new - output within 2e-2 of the output range, every gradient with cosine >= 0.999 to the
    fp32-row form and no further from the fp32 head than 1.5x the fp32-row form is (+1e-3 of the gradient's range)."""
    from catre_amd import train_ops as T

    g = torch.Generator().manual_seed(41 + B)
    P = N + M
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    t = dict(x=r(B * P, 64), w0=r(256, 64, 1, sc=0.1), bias=r(2 * B, 256, sc=0.3), g0=1 + r(256, sc=0.1), be0=r(256, sc=0.1),
             w1=r(256, 256, 1, sc=0.06), b1=r(256, sc=0.1), g1=1 + r(256, sc=0.1), be1=r(256, sc=0.1), wn=r(3, 256, sc=0.1),
             bn=r(3, sc=0.1), wp=r(1, P, 1, sc=0.05), bp=r(1, sc=0.1))
    for v in t.values():
        v.requires_grad_(True)
    t["dout"] = r(B, 3)

    def rows_bf16():
        return T.rot_head_lp(t["x"], t["w0"], t["bias"], t["g0"], t["be0"], t["w1"], t["b1"], t["g1"], t["be1"], t["wn"], t["bn"],
                             t["wp"], t["bp"], B, N, M)

    def rows_fp32():
        a = T.rot_l0_block(t["x"], t["w0"], t["bias"], t["g0"], t["be0"], B, N, M)
        return T.rot_l1_tail_lp(a, t["w1"], t["b1"], t["g1"], t["be1"], t["wn"], t["bn"], t["wp"], t["bp"], B, N, M)

    def fp32_head():
        a = T.rot_l0_block(t["x"], t["w0"], t["bias"], t["g0"], t["be0"], B, N, M)
        y3 = T.rot_l1_block(a, t["w1"], t["b1"], t["g1"], t["be1"], t["wn"], t["bn"], B, N, M)
        return T.weighted_point_sum(y3, t["wp"], t["bp"], B, P)

    with T.amp_mode("bf16"):
        assert T.rot_head_lp_ok(t["x"], t["w0"], t["w1"], t["b1"], N, M)
        oh, gh = _run(rows_bf16, t)
        of, gf = _run(rows_fp32, t)
    with T.amp_mode("fp32"):
        o32, g32 = _run(fp32_head, t)
    rng = float(o32.abs().max())
    assert float((oh - of).abs().max()) <= 2e-2 * rng
    for k in g32:
        a, b, c = gh[k].reshape(-1), gf[k].reshape(-1), g32[k].reshape(-1)
        if float(c.norm()) == 0.0:
            assert float(a.norm()) == 0.0, k
            continue
        cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
        assert cos >= 0.999, (k, cos)
        scale = float(c.abs().max())
        eh, ef = float((a - c).abs().max()) / scale, float((b - c).abs().max()) / scale
        assert eh <= 1.5 * ef + 1e-3, (k, eh, ef)


def test_autocast_iteration_with_bf16_rows_tracks_the_fp32_iteration():
    """The default autocast training iteration (heads with bf16 rows) against the fp32 iteration, with the criteria of
    test_hip_train.test_amp_training_iteration_tracks_fp32, and against the fp32-row form of the same iteration."""
    from catre_amd import train_ops as T
    from test_hip_train import _train_setup, _iteration

    B, N, M = 6, 256, 192
    cfg, sd, kw, sym, ((model, opt),) = _train_setup(B, N, M, 33, 1)

    def run(autocast, rows):
        model.cfg.MODEL.CATRE.TRAIN_KERNELS = dict(lp_rot_bf16_rows=rows)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            ld = _iteration(model, kw, sym)
        return ld, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    l32, g32 = run(False, True)
    lh, gh = run(True, True)
    lf, gf = run(True, False)
    for k in l32:
        assert abs(float(lh[k]) - float(l32[k])) <= 3e-2 * abs(float(l32[k])) + 1e-3, (k, float(lh[k]), float(l32[k]))
    checked = 0
    for k, g in g32.items():
        if g.numel() < 4096 or float(g.norm()) < 1e-8:
            continue
        cos = float(torch.nn.functional.cosine_similarity(g.reshape(-1), gh[k].reshape(-1), dim=0))
        ratio = float(gh[k].norm() / g.norm())
        assert cos >= 0.98 and 0.9 <= ratio <= 1.1, (k, cos, ratio)
        cos_f = float(torch.nn.functional.cosine_similarity(gf[k].reshape(-1), gh[k].reshape(-1), dim=0))
        assert cos_f >= 0.995, (k, cos_f)
        checked += 1
    assert checked >= 20


def test_autocast_iteration_with_differentiable_points_takes_the_layerwise_encoder():
    """The autocast fused encoder forward saves bf16 activation rows that only the row-sparse chains read; with a
    differentiable 3-d input the first STN stack is not a chain (train_forward.fused_lp_ok), so the whole encoder runs
    layer-wise: the parameter gradients track the default autocast iteration's and the points get a finite gradient."""
    from test_hip_train import _train_setup, _iteration

    B, N, M = 4, 128, 128
    cfg, sd, kw, sym, ((model, opt),) = _train_setup(B, N, M, 35, 1)

    def run(diff_points):
        kk = dict(kw)
        if diff_points:
            kk["x"] = kk["x"].clone().requires_grad_(True)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ld = _iteration(model, kk, sym)
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        return ld, grads, (kk["x"].grad if diff_points else None)

    l0, g0, _ = run(False)
    l1, g1, gx = run(True)
    assert gx is not None and torch.isfinite(gx).all() and float(gx.abs().max()) > 0
    for k in l0:
        assert abs(float(l1[k]) - float(l0[k])) <= 3e-2 * abs(float(l0[k])) + 1e-3, (k, float(l1[k]), float(l0[k]))
    checked = 0
    for k, g in g0.items():
        if g.numel() < 4096 or float(g.norm()) < 1e-8:
            continue
        cos = float(torch.nn.functional.cosine_similarity(g.reshape(-1), g1[k].reshape(-1), dim=0))
        assert cos >= 0.97, (k, cos)
        checked += 1
    assert checked >= 20


def test_groupnorm0_inside_the_second_linears_staging_changes_nothing():
    """catre_op_gn_gelu_gemm_rows_h (GroupNorm-0 + GELU applied while the 256 -> 256 GEMM stages its operand) against the
    separate pass + GEMM: the head's output and every gradient bit for bit."""
    from catre_amd import train_ops as T

    B, N, M = 4, 256, 320
    g = torch.Generator().manual_seed(77)
    P = N + M
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    t = dict(x=r(B * P, 64), w0=r(256, 64, 1, sc=0.1), bias=r(2 * B, 256, sc=0.3), g0=1 + r(256, sc=0.1), be0=r(256, sc=0.1),
             w1=r(256, 256, 1, sc=0.06), b1=r(256, sc=0.1), g1=1 + r(256, sc=0.1), be1=r(256, sc=0.1), wn=r(3, 256, sc=0.1),
             bn=r(3, sc=0.1), wp=r(1, P, 1, sc=0.05), bp=r(1, sc=0.1))
    for v in t.values():
        v.requires_grad_(True)
    t["dout"] = r(B, 3)
    fn = lambda: T.rot_head_lp(t["x"], t["w0"], t["bias"], t["g0"], t["be0"], t["w1"], t["b1"], t["g1"], t["be1"], t["wn"],
                               t["bn"], t["wp"], t["bp"], B, N, M)
    res = []
    with T.amp_mode("bf16"):
        for fused in (True, False):
            with T.train_kernels(lp_rot_fuse_gn0=fused):
                res.append(_run(fn, t))
    (o1, g1), (o2, g2) = res
    assert torch.equal(o1, o2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


def test_autocast_iteration_off_the_64_point_grid_takes_the_layerwise_heads():
    """N, M not multiples of 64: the one-node heads do not apply (rot_head_lp_ok), the autocast iteration runs on the
    layer-wise ops and still tracks the fp32 iteration."""
    from catre_amd import train_ops as T
    from test_hip_train import _train_setup, _iteration

    B, N, M = 5, 96, 40
    cfg, sd, kw, sym, ((model, opt),) = _train_setup(B, N, M, 35, 1)

    def run(autocast):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            ld = _iteration(model, kw, sym)
        return ld, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    l32, g32 = run(False)
    lh, gh = run(True)
    for k in l32:
        assert abs(float(lh[k]) - float(l32[k])) <= 3e-2 * abs(float(l32[k])) + 1e-3, (k, float(lh[k]), float(l32[k]))
    checked = 0
    for k, g in g32.items():
        assert torch.isfinite(gh[k]).all(), k
        if g.numel() < 4096 or float(g.norm()) < 1e-8:
            continue
        cos = float(torch.nn.functional.cosine_similarity(g.reshape(-1), gh[k].reshape(-1), dim=0))
        assert cos >= 0.97, (k, cos)
        checked += 1
    assert checked >= 20


def test_split_iteration_with_one_pass_head_backward_matches_the_layerwise_split_backward():
    """COMPUTE_DTYPE='split': the rot heads' blocks as one-pass nodes (k_rot_l0_bwd on the fp32 pipe, k_rot_l1_bwd_sp with
    hi + lo operands) against the layer-wise split dgrad / wgrad GEMMs: identical losses (the forward is the same fused
    kernel), every gradient within 2e-4 of its largest entry - both are fp32-grade - and within the split mode's bar (5e-2
    relative L2) of the fp32 iteration."""
    from catre_amd import train_forward as F
    from catre_amd import train_ops as T
    from test_hip_train import _train_setup, _iteration

    B, N, M = 6, 256, 192
    cfg, sd, kw, sym, ((model, opt),) = _train_setup(B, N, M, 37, 1)

    def run(mode, one_pass):
        model.cfg.MODEL.CATRE.TRAIN_KERNELS = dict(split_l0_one_pass=one_pass, split_l1_one_pass=one_pass)
        opt.zero_grad(set_to_none=True)
        with T.amp_mode(mode):
            ld = _iteration(model, kw, sym)
        return ld, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    l1, g1 = run("split", True)
    l2, g2 = run("split", False)
    l32, g32 = run("fp32", True)
    for k in l2:
        assert torch.equal(l1[k], l2[k]), k
    for k in g2:
        scale = float(g32[k].abs().max()) + 1e-30
        assert float((g1[k] - g2[k]).abs().max()) / scale <= 2e-4, (k, float((g1[k] - g2[k]).abs().max()) / scale)
        # (against the fp32 iteration: the bar of test_hip_train's split test, 5e-2 relative L2 - max-pool winners flip)
        assert float((g1[k] - g32[k]).norm() / (g32[k].norm() + 1e-30)) <= 5e-2, k


@pytest.mark.parametrize("mode", ["bf16", "split"])
def test_one_pass_heads_at_a_batch_whose_workgroups_walk_several_tiles(mode):
    """B = 40 objects of N = M = 1024 points: the one-pass backward kernels run 6 chunks of 6 tiles per object (the small
    cases above give every workgroup one tile, the bench shape one workgroup per object) - against the layer-wise backward
    of the same mode."""
    from catre_amd import train_forward as F
    from catre_amd import train_ops as T
    from test_hip_train import _train_setup, _iteration

    B, N, M = 40, 1024, 1024
    cfg, sd, kw, sym, ((model, opt),) = _train_setup(B, N, M, 39, 1)

    def run(one_pass):
        # fp32 rows: the layer-wise path's own tensors, so the comparison is tight
        model.cfg.MODEL.CATRE.TRAIN_KERNELS = dict(split_l0_one_pass=one_pass, split_l1_one_pass=one_pass,
                                                   fused_lp_rot=one_pass, lp_rot_bf16_rows=False)
        opt.zero_grad(set_to_none=True)
        with T.amp_mode(mode):
            ld = _iteration(model, kw, sym)
        return ld, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    l1, g1 = run(True)
    l2, g2 = run(False)
    for k in l2:
        assert torch.equal(l1[k], l2[k]), k
    for k in g2:
        scale = float(g2[k].abs().max()) + 1e-30
        err = float((g1[k] - g2[k]).abs().max()) / scale
        assert err <= (2e-4 if mode == "split" else 2e-3), (k, err)
