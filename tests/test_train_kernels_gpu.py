"""Backward building blocks of the training step (conv weight / data gradient, training-mode BatchNorm) against PyTorch
CPU autograd of the same layer -- the reference's training step is plain autograd over nn.Conv2d / nn.BatchNorm2d
(train.py:119-131).  f32: relative to the gradient's max 2e-5 (reduction order differs); 16-bit: inputs are quantised the
same way on both sides, products are exact in f32, so the same bound applies to the weight gradient; outputs stored in
16 bits carry their storage rounding."""
import math

import pytest
import torch
import torch.nn.functional as F

from cfpnet_amd import hip, ops, train_ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
OUT_TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2, torch.float16: 1.5e-3}

CASES = [  # B, H, W, Cin, Cout, k, stride, pads (t, l, b, r)
    (2, 12, 16, 16, 24, 3, 1, (1, 1, 1, 1)),
    (2, 9, 11, 8, 40, 3, 2, (0, 0, 1, 1)),        # stem-like: TF SAME on odd sizes
    (1, 10, 14, 16, 64, 3, 2, (0, 0, 1, 1)),
    (3, 8, 8, 136, 816, 1, 1, (0, 0, 0, 0)),      # pointwise, ragged Cout tile
    (1, 30, 40, 32, 32, 6, 6, (0, 0, 0, 0)),      # GSA sr conv: kernel = stride
    (2, 33, 47, 8, 16, 3, 1, (1, 1, 1, 1)),       # many rows, tiny channels
    (1, 1, 300, 64, 128, 1, 1, (0, 0, 0, 0)),     # Linear
    (2, 128, 160, 16, 128, 3, 1, (1, 1, 1, 1)),   # 40 960 rows x 128 output channels (the decoder / head class of weight gradients)
    (1, 200, 180, 24, 136, 1, 1, (0, 0, 0, 0)),   # ... pointwise, second channel tile 8 wide
    (1, 192, 192, 8, 256, 3, 1, (1, 1, 1, 1)),    # ... two full channel tiles, K = 72 (ragged K tile)
    (1, 192, 192, 32, 128, 3, 1, (1, 1, 1, 1)),   # K = 288
    (1, 200, 180, 40, 136, 3, 2, (0, 0, 1, 1)),   # ... strided, ragged in both tile directions (Cout 136, K 360)
    (1, 200, 184, 40, 136, 3, 1, (1, 1, 1, 1)),   # ... 36 800 rows, ragged tiles
]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def nhwc(t):
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def _ref(case, dtype):
    B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
    x = rnd(B, Cin, H, W, seed=1).to(dtype).float().requires_grad_(True)
    w = rnd(Cout, Cin, k, k, seed=2, scale=1.0 / math.sqrt(Cin * k * k)).to(dtype).float().requires_grad_(True)
    y = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, None, s)
    dy = rnd(*y.shape, seed=3).to(dtype).float()
    y.backward(dy)
    return x.detach(), w.detach(), dy, x.grad, w.grad, y.shape[2], y.shape[3]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES)
def test_conv_weight_gradient(case, dtype):
    B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
    x, w, dy, dx_ref, dw_ref, Ho, Wo = _ref(case, dtype)
    dw = train_ops.conv2d_wgrad(nhwc(x).to(dtype).to(DEV), nhwc(dy).to(dtype).to(DEV), B, H, W, k, k, s, pt, pl, Ho, Wo)
    torch.cuda.synchronize()
    got = dw.cpu().reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    err = float((got - dw_ref).abs().max()) / float(dw_ref.abs().max())
    assert err < 2e-5, (case, dtype, err)
    # accumulate form: dw <- 0.5 * dw + grad, and bit-reproducible
    dw2 = train_ops.conv2d_wgrad(nhwc(x).to(dtype).to(DEV), nhwc(dy).to(dtype).to(DEV), B, H, W, k, k, s, pt, pl, Ho, Wo, dw=dw.clone(), beta=0.5)
    dw3 = train_ops.conv2d_wgrad(nhwc(x).to(dtype).to(DEV), nhwc(dy).to(dtype).to(DEV), B, H, W, k, k, s, pt, pl, Ho, Wo)
    torch.cuda.synchronize()
    assert torch.equal(dw3, dw)
    assert float((dw2 - 1.5 * dw).abs().max()) <= 1e-6 * float(dw.abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES)
def test_conv_data_gradient(case, dtype):
    B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
    x, w, dy, dx_ref, dw_ref, Ho, Wo = _ref(case, dtype)
    w2d = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dtype).to(DEV)
    wt = train_ops.conv2d_weight_flip(w2d, Cout, k, k, Cin)
    want_wt = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, k * k * Cout).to(dtype)
    assert torch.equal(wt.cpu(), want_wt)
    dyd = nhwc(dy).to(dtype).to(DEV)
    dx = train_ops.conv2d_dgrad(dyd, wt, B, H, W, Cin, k, k, s, pt, pl, Ho, Wo)
    torch.cuda.synchronize()
    got = dx.float().cpu().reshape(B, H, W, Cin).permute(0, 3, 1, 2)
    scale = float(dx_ref.abs().max())
    assert float((got - dx_ref).abs().max()) <= OUT_TOL[dtype] * scale, (case, dtype)
    # accumulate into an existing gradient (skip connections)
    base = rnd(B * H * W, Cin, seed=9).to(dtype).to(DEV)
    acc = train_ops.conv2d_dgrad(dyd, wt, B, H, W, Cin, k, k, s, pt, pl, Ho, Wo, dx=base.clone(), accumulate=True)
    torch.cuda.synchronize()
    assert float((acc.float() - (base.float() + dx.float())).abs().max()) <= 2 * OUT_TOL[dtype] * (scale + 4.0)


ACTS = [hip.ACT_NONE, hip.ACT_RELU, hip.ACT_LRELU, hip.ACT_SILU, hip.ACT_GELU, hip.ACT_SIGMOID]
TORCH_ACT = {hip.ACT_NONE: lambda t: t, hip.ACT_RELU: F.relu, hip.ACT_LRELU: lambda t: F.leaky_relu(t, 0.01), hip.ACT_SILU: F.silu,
             hip.ACT_GELU: F.gelu, hip.ACT_SIGMOID: torch.sigmoid}


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("rows,C", [(2 * 30 * 40, 136), (5000, 8), (77, 1392), (3 * 60 * 80, 64)])
def test_batchnorm_training_forward_backward(rows, C, act, dtype):
    eps, mom = 1e-3, 0.1
    x = (rnd(rows, C, seed=1) * 1.7 + 0.3).to(dtype).float().requires_grad_(True)
    gamma = (rnd(C, seed=2).abs() + 0.5).requires_grad_(True)
    beta = rnd(C, seed=3).requires_grad_(True)
    rm, rv = rnd(C, seed=4), rnd(C, seed=5).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = TORCH_ACT[act](F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, mom, eps))
    dy = rnd(rows, C, seed=6).to(dtype).float()
    y.backward(dy)

    bn = train_ops.BatchNormTrain(C, DEV, eps=eps, momentum=mom)
    xd, rmd, rvd = x.detach().to(dtype).to(DEV), rm.to(DEV), rv.to(DEV)
    yd = bn.forward(xd, gamma.detach().to(DEV), beta.detach().to(DEV), rmd, rvd, act)
    dx, dgamma, dbeta = bn.backward(xd, dy.to(dtype).to(DEV), act)
    torch.cuda.synchronize()
    tol = OUT_TOL[dtype]
    assert float((bn.mean.cpu() - x.detach().mean(0)).abs().max()) < 1e-5
    assert float((bn.var.cpu() - x.detach().var(0, unbiased=False)).abs().max()) < 2e-5 * float(x.detach().var(0).max())
    assert torch.allclose(rmd.cpu(), rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rvd.cpu(), rv_ref, rtol=1e-5, atol=1e-6)
    assert float((yd.float().cpu() - y.detach()).abs().max()) <= tol * float(y.detach().abs().max()) + 1e-6
    gs = float(x.grad.abs().max())
    assert float((dx.float().cpu() - x.grad).abs().max()) <= tol * gs + 2e-5 * gs
    assert float((dgamma.cpu() - gamma.grad).abs().max()) <= 3e-5 * float(gamma.grad.abs().max()) + 1e-5
    assert float((dbeta.cpu() - beta.grad).abs().max()) <= 3e-5 * float(beta.grad.abs().max()) + 1e-5


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
@pytest.mark.parametrize("via_partials", [False, True])
def test_batchnorm_running_statistics_survive_a_non_finite_batch(bad, via_partials):
    """An activation overflow in an fp16 forward (the step cfp_grad_clip_factor then makes the optimizer skip) must not reach the
    running statistics: they outlive the step and every later validation engine folds them.  The poisoned channel keeps its running
    mean / variance, the clean channels are updated as usual -- through the statistics pass and through the conv-epilogue partials."""
    rows, C, mom = 512, 16, 0.1
    x = rnd(rows, C, seed=1)
    x[7, 3] = bad
    rm, rv = rnd(C, seed=4), rnd(C, seed=5).abs() + 0.5
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    bn = train_ops.BatchNormTrain(C, DEV, eps=1e-3, momentum=mom)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    if via_partials:      # two row tiles of 256 rows: (mean, M2) per tile as the conv epilogue leaves them
        xt = x.reshape(2, 256, C)
        m = xt.mean(1)
        part = torch.stack([m, ((xt - m[:, None]) ** 2).sum(1)], 1).contiguous().to(DEV)      # [2][2][C]
        bn.forward(x.to(DEV), g, b, rmd, rvd, hip.ACT_NONE, mom=(part, 2, 256))
    else:
        bn.forward(x.to(DEV), g, b, rmd, rvd, hip.ACT_NONE)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(rmd).all()) and bool(torch.isfinite(rvd).all())
    assert float(rmd[3]) == float(rm[3]) and float(rvd[3]) == float(rv[3])
    keep = [c for c in range(C) if c != 3]
    want_m = (1 - mom) * rm + mom * x.mean(0)
    want_v = (1 - mom) * rv + mom * x.var(0, unbiased=True)
    assert torch.allclose(rmd.cpu()[keep], want_m[keep], rtol=1e-5, atol=1e-6) and torch.allclose(rvd.cpu()[keep], want_v[keep], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 30, 40, 112, 448, 1, 1), (2, 60, 80, 40, 160, 3, 1), (1, 120, 160, 8, 40, 3, 2), (3, 13, 17, 136, 816, 1, 1),
                                  (1, 9, 11, 232, 1392, 1, 1), (16, 26, 34, 448, 112, 1, 1), (2, 104, 136, 16, 16, 3, 1), (1, 1, 70, 8, 32, 1, 1)])
def test_conv_epilogue_moments_feed_batchnorm(case, dtype):
    """cfp_conv2d_nhwc_moments: the conv kernel leaves per-row-tile (mean, M2) of its STORED output; BatchNorm merged from them must equal
    BatchNorm with its own statistics pass over that output (same values, another exact merge order) and torch's batch statistics of it.
    Data with a large common offset (mean >> std): the tile sums are shifted, so nothing cancels."""
    B, H, W, Cin, Cout, k, s = case
    pt = pl = (k - 1) // 2 if s == 1 else 0
    Ho, Wo = (H + 2 * pt - k) // s + 1, (W + 2 * pl - k) // s + 1
    x = (rnd(B * H * W, Cin, seed=1) + 3.0).to(dtype).to(DEV)
    w = (rnd(Cout, k * k * Cin, seed=2) / math.sqrt(k * k * Cin)).to(dtype).to(DEV)
    bias = (rnd(Cout, seed=3) * 4.0).to(DEV)
    M = B * Ho * Wo
    y1, y2 = torch.empty(M, Cout, dtype=dtype, device=DEV), torch.empty(M, Cout, dtype=dtype, device=DEV)
    A = lambda t: ops.Act(t, 0, t.shape[1])
    mom, ns, rps = ops.conv2d_moments(A(x), w, bias, A(y1), B, H, W, k, k, s, pt, pl, Ho, Wo)
    ops.conv2d(A(x), w, None, bias, A(y2), B, H, W, k, k, s, pt, pl, Ho, Wo)
    torch.cuda.synchronize()
    assert ns > 0 and (ns - 1) * rps < M <= ns * rps, (ns, rps, M)
    assert torch.equal(y1.view(torch.int16), y2.view(torch.int16)), "the moments epilogue must not change the output"
    gamma, beta = (rnd(Cout, seed=4).abs() + 0.5).to(DEV), rnd(Cout, seed=5).to(DEV)
    res = []
    for m in ((mom, ns, rps), None):
        bn = train_ops.BatchNormTrain(Cout, DEV, eps=1e-3, momentum=0.1)
        rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
        out = bn.forward(y1, gamma, beta, rm, rv, hip.ACT_SILU, mom=m)
        torch.cuda.synchronize()
        res.append((bn.mean.clone(), bn.var.clone(), rm, rv, out.float()))
    yf = y1.double()
    mean64, var64 = yf.mean(0), yf.var(0, unbiased=False)
    for mean, var, rm, rv, out in res:
        assert float((mean.double() - mean64).abs().max()) <= 2e-6 * float(mean64.abs().max()) + 1e-6
        assert float(((var.double() - var64).abs() / var64.clamp(min=1e-9)).max()) <= 2e-4
    assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-6 * float(mean64.abs().max()) + 1e-6
    assert float((res[0][4] - res[1][4]).abs().max()) <= {torch.bfloat16: 2.0 ** -6, torch.float16: 2.0 ** -9}[dtype] * float(res[1][4].abs().max())
    assert torch.allclose(res[0][3], res[1][3], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("rows,C", [(20000, 16), (333, 264), (1, 8), (200000, 8)])
def test_batchnorm_statistics_one_pass_is_stable(rows, C):
    """The statistics are taken in one pass over x (shifted sums per thread + exact merging of (n, mean, M2) triples): data whose
    mean is 2*10^4 standard deviations away from zero -- where E[x^2] - E[x]^2 in float32 returns noise -- plus a few far
    outliers must still give the float64 mean / variance."""
    g = torch.Generator().manual_seed(3)
    x = (1000.0 + 0.05 * torch.randn(rows, C, generator=g)).float()
    if rows > 100:
        x[7, :] += 3.0
        x[rows // 2, ::2] -= 5.0
    x64 = x.double()
    bn = train_ops.BatchNormTrain(C, DEV, eps=1e-5, momentum=0.1)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    y = bn.forward(x.to(DEV), torch.ones(C, device=DEV), torch.zeros(C, device=DEV), rm, rv, hip.ACT_NONE)
    torch.cuda.synchronize()
    mean64, var64 = x64.mean(0), x64.var(0, unbiased=False)
    assert float((bn.mean.cpu().double() - mean64).abs().max()) <= 2e-7 * 1000.0
    assert float(((bn.var.cpu().double() - var64).abs() / var64.clamp(min=1e-12)).max()) <= (2e-3 if rows > 1 else 1.0)
    want = (x64 - mean64) / (var64 + 1e-5).sqrt()
    assert float((y.cpu().double() - want).abs().max()) <= 2e-3 * float(want.abs().max()) + 1e-3
    unb = rows / max(rows - 1, 1)
    assert torch.allclose(rv.cpu().double(), 0.9 + 0.1 * var64 * unb, rtol=2e-3, atol=1e-6) and torch.allclose(rm.cpu().double(), 0.1 * mean64, rtol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(4800, 128), (257, 32), (19200, 64), (1, 256)])
def test_layernorm_backward(rows, C, dtype):
    x = (rnd(rows, C, seed=1) * 1.3 + 0.2).to(dtype).float().requires_grad_(True)
    gamma = (rnd(C, seed=2).abs() + 0.5).requires_grad_(True)
    beta = rnd(C, seed=3).requires_grad_(True)
    dy = rnd(rows, C, seed=4).to(dtype).float()
    F.layer_norm(x, (C,), gamma, beta, 1e-5).backward(dy)
    xd, dyd = x.detach().to(dtype).to(DEV), dy.to(dtype).to(DEV)
    dx, dg, db = train_ops.layernorm_bwd(xd, dyd, gamma.detach().to(DEV), 1e-5)
    base = rnd(rows, C, seed=5).to(dtype).to(DEV)
    dx2, _, _ = train_ops.layernorm_bwd(xd, dyd, gamma.detach().to(DEV), 1e-5, dx=base.clone(), accumulate=True)
    torch.cuda.synchronize()
    gs = float(x.grad.abs().max())
    assert float((dx.float().cpu() - x.grad).abs().max()) <= (OUT_TOL[dtype] + 2e-5) * gs
    assert float((dx2.float() - (base.float() + dx.float())).abs().max()) <= 2 * OUT_TOL[dtype] * (gs + 4.0) + 1e-5
    assert float((dg.cpu() - gamma.grad).abs().max()) <= 3e-5 * float(gamma.grad.abs().max()) + 1e-5
    assert float((db.cpu() - beta.grad).abs().max()) <= 3e-5 * float(beta.grad.abs().max()) + 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", ACTS)
def test_activation_backward_and_bias_gradient(act, dtype):
    rows, C = 3001, 72
    z = rnd(rows, C, seed=1, scale=2.0).to(dtype).float().requires_grad_(True)
    dy = rnd(rows, C, seed=2).to(dtype).float()
    TORCH_ACT[act](z).backward(dy)
    dz = train_ops.act_bwd(z.detach().to(dtype).to(DEV), dy.to(dtype).to(DEV), act)
    bias_grad = train_ops.colsum(dy.to(dtype).to(DEV))
    torch.cuda.synchronize()
    assert float((dz.float().cpu() - z.grad).abs().max()) <= (OUT_TOL[dtype] + 1e-5) * float(z.grad.abs().max())
    assert float((bias_grad.cpu() - dy.sum(0)).abs().max()) <= 2e-5 * float(dy.sum(0).abs().max()) + 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 30, 40, 224, 2, (0, 0, 1, 1)), (1, 15, 20, 1392, 1, (1, 1, 1, 1)), (2, 9, 7, 64, 2, (1, 1, 1, 1)),
                                  (3, 16, 16, 16, 1, (1, 1, 1, 1)), (1, 31, 33, 8, 2, (0, 0, 1, 1))])
def test_depthwise3x3_backward(case, dtype):
    B, H, W, C, s, (pt, pl, pb, pr) = case
    x = rnd(B, C, H, W, seed=1).to(dtype).float().requires_grad_(True)
    w = rnd(C, 1, 3, 3, seed=2, scale=0.4).to(dtype).float().requires_grad_(True)
    y = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, None, s, 0, 1, C)
    Ho, Wo = y.shape[2:]
    dy = rnd(*y.shape, seed=3).to(dtype).float()
    y.backward(dy)
    w9c = w.detach().reshape(C, 9).t().contiguous().to(dtype).to(DEV)
    dyd, xd = nhwc(dy).to(dtype).to(DEV), nhwc(x.detach()).to(dtype).to(DEV)
    dx = train_ops.dwconv3x3_dgrad(dyd, w9c, B, H, W, s, pt, pl, Ho, Wo)
    dw = train_ops.dwconv3x3_wgrad(xd, dyd, B, H, W, s, pt, pl, Ho, Wo)
    dw2 = train_ops.dwconv3x3_wgrad(xd, dyd, B, H, W, s, pt, pl, Ho, Wo, dw=dw.clone(), beta=1.0)
    torch.cuda.synchronize()
    got = dx.float().cpu().reshape(B, H, W, C).permute(0, 3, 1, 2)
    assert float((got - x.grad).abs().max()) <= (OUT_TOL[dtype] + 1e-5) * float(x.grad.abs().max())
    want_dw = w.grad.reshape(C, 9).t()
    assert float((dw.cpu() - want_dw).abs().max()) <= 3e-5 * float(want_dw.abs().max())
    assert float((dw2 - 2 * dw).abs().max()) <= 1e-6 * float(dw.abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
def test_squeeze_excite_and_small_backward_pieces(dtype):
    B, HW, C = 3, 300, 136
    x = rnd(B * HW, C, seed=1).to(dtype)
    dy = rnd(B * HW, C, seed=2).to(dtype)
    gate, add = torch.rand(B, C, generator=torch.Generator().manual_seed(3)), rnd(B, C, seed=4, scale=0.01)
    xd, dyd = x.to(DEV), dy.to(DEV)
    dot = train_ops.channel_dot(xd, dyd, B, HW)
    dx = train_ops.bcast_fma(dyd, gate.to(DEV), add.to(DEV), B, HW)
    z = train_ops.axpby(xd, dyd, 0.5, -2.0)
    z1 = train_ops.axpby(xd, None, 3.0, 0.0)
    H, W, Wt = 15, 20, 40
    tab = torch.zeros(30 * Wt, C, device=DEV)
    train_ops.rowtable_grad(dyd, tab, B, H, W, Wt, 3, 7)
    torch.cuda.synchronize()
    xf, dyf = x.float(), dy.float()
    want = (xf * dyf).reshape(B, HW, C).sum(1)
    assert float((dot.cpu() - want).abs().max()) <= 3e-5 * float(want.abs().max())
    want_dx = dyf.reshape(B, HW, C) * gate[:, None] + add[:, None]
    assert float((dx.float().cpu().reshape(B, HW, C) - want_dx).abs().max()) <= OUT_TOL[dtype] * float(want_dx.abs().max()) + 1e-6
    assert float((z.float().cpu() - (0.5 * xf - 2.0 * dyf)).abs().max()) <= OUT_TOL[dtype] * 10 + 1e-6
    assert float((z1.float().cpu() - 3.0 * xf).abs().max()) <= OUT_TOL[dtype] * 12 + 1e-6
    want_tab = torch.zeros(30, Wt, C)
    want_tab[3:3 + H, 7:7 + W] = dyf.reshape(B, H, W, C).sum(0)
    assert float((tab.cpu().reshape(30, Wt, C) - want_tab).abs().max()) <= 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_index_rows_and_its_adjoint(dtype):
    n_src, n_out, C = 5000, 4200, 64
    g = torch.Generator().manual_seed(1)
    perm = torch.randperm(n_src, generator=g)[:n_out].to(torch.int32)
    perm[::7] = -1                                            # padding rows
    x, gy = rnd(n_src, C, seed=2).to(dtype), rnd(n_out, C, seed=3).to(dtype)
    idx = perm.to(DEV)
    y = train_ops.index_rows(x.to(DEV), idx)
    inv = train_ops.inverse_index(idx, n_src)
    gx = train_ops.index_rows(gy.to(DEV), inv)
    torch.cuda.synchronize()
    want = torch.where((perm >= 0)[:, None], x[perm.clamp(min=0).long()], torch.zeros(()).to(dtype))
    assert torch.equal(y.cpu(), want)
    # <gather(x), gy> == <x, gather_inv(gy)>
    lhs = float((y.float().cpu() * gy.float()).sum()); rhs = float((x.float() * gx.float().cpu()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Hs,Ws,Hd,Wd,C", [(2, 15, 20, 30, 40, 64), (1, 28, 28, 32, 32, 32), (2, 32, 32, 28, 28, 16), (1, 7, 9, 7, 9, 8), (1, 1, 5, 4, 13, 8)])
def test_bilinear_resize_backward(B, Hs, Ws, Hd, Wd, C, dtype):
    x = rnd(B, C, Hs, Ws, seed=1).to(dtype).float().requires_grad_(True)
    y = F.interpolate(x, (Hd, Wd), mode="bilinear", align_corners=True)
    dy = rnd(*y.shape, seed=2).to(dtype).float()
    y.backward(dy)
    dx = train_ops.resize_bilinear_bwd(nhwc(dy).to(dtype).to(DEV), B, Hs, Ws, Hd, Wd)
    torch.cuda.synchronize()
    got = dx.float().cpu().reshape(B, Hs, Ws, C).permute(0, 3, 1, 2)
    assert float((got - x.grad).abs().max()) <= (OUT_TOL[dtype] + 2e-5) * float(x.grad.abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW", [(2, 208 * 272 // 16), (3, 77)])
def test_bin_head_forward_backward(B, HW, dtype):
    NB = 256
    logits = rnd(B * HW, NB, seed=1, scale=2.0).to(dtype).float().requires_grad_(True)
    wn = torch.softmax(rnd(B, NB, seed=2), 1).requires_grad_(True)
    widths = F.pad((10.0 - 1e-3) * wn, (1, 0), value=1e-3)
    edges = torch.cumsum(widths, 1)
    centers = 0.5 * (edges[:, :-1] + edges[:, 1:])
    pred = (torch.softmax(logits.reshape(B, HW, NB), 2) * centers[:, None]).sum(2)
    dpred = rnd(B, HW, seed=3)
    pred.backward(dpred)
    e_d, c_d = train_ops.bin_centers(wn.detach().to(DEV), 1e-3, 10.0)
    ld = logits.detach().to(dtype).to(DEV)
    p_d = train_ops.softmax_expect(ld, c_d, B, HW)
    dl, dc = train_ops.softmax_expect(ld, c_d, B, HW, dpred=dpred.reshape(-1).to(DEV))
    dwn = train_ops.bin_centers_bwd(dc, 1e-3, 10.0)
    torch.cuda.synchronize()
    assert torch.allclose(e_d.cpu(), edges.detach(), rtol=1e-5, atol=1e-5) and torch.allclose(c_d.cpu(), centers.detach(), rtol=1e-5, atol=1e-5)
    assert float((p_d.cpu().reshape(B, HW) - pred.detach()).abs().max()) <= 2e-5 * 10
    assert float((dl.float().cpu() - logits.grad).abs().max()) <= (OUT_TOL[dtype] + 2e-5) * float(logits.grad.abs().max())
    assert float((dwn.cpu() - wn.grad).abs().max()) <= 5e-5 * float(wn.grad.abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,L,S,heads,d", [(6, 49, 16, 4, 16), (3, 144, 144, 8, 4), (2, 1200, 30, 8, 16), (5, 36, 36, 8, 16), (2, 300, 130, 4, 32),
                                           (4, 196, 16, 4, 8), (2, 884, 884, 4, 32), (2, 3536, 500, 4, 16), (130, 36, 16, 4, 8), (3, 16, 16, 4, 32), (2, 64, 64, 8, 4)])
def test_linear_attention_forward_backward(N, L, S, heads, d, dtype):
    from oracle import cfpnet_oracle as O
    q = rnd(N, L, heads, d, seed=1).to(dtype).float().requires_grad_(True)
    k = rnd(N, S, heads, d, seed=2).to(dtype).float().requires_grad_(True)
    v = rnd(N, S, heads, d, seed=3).to(dtype).float().requires_grad_(True)
    out = O.linear_attention(q, k, v)
    dout = rnd(*out.shape, seed=4).to(dtype).float()
    out.backward(dout)
    f = lambda t, n: t.detach().reshape(N * n, heads * d).to(dtype).to(DEV)
    qd, kd, vd = f(q, L), f(k, S), f(v, S)
    o_d, state = train_ops.linattn_fwd(qd, kd, vd, N, L, S, heads, d)
    dq, dk, dv = train_ops.linattn_bwd(qd, kd, vd, dout.reshape(N * L, heads * d).to(dtype).to(DEV), state, N, L, S, heads, d)
    o2, _ = train_ops.linattn_fwd(qd, kd, vd, N, L, S, heads, d)
    torch.cuda.synchronize()
    assert torch.equal(o_d, o2)                                    # fixed summation order
    tol = OUT_TOL[dtype] + 3e-5
    chk = lambda got, want, n: float((got.float().cpu().reshape(N, n, heads, d) - want).abs().max()) <= tol * float(want.abs().max())
    assert chk(o_d, out.detach(), L) and chk(dq, q.grad, L) and chk(dk, k.grad, S) and chk(dv, v.grad, S)
    # the token-split launch (few groups) against one workgroup per (group, head): same values up to f32 summation order
    o1, st1 = train_ops.linattn_fwd(qd, kd, vd, N, L, S, heads, d, split=False)
    g1 = train_ops.linattn_bwd(qd, kd, vd, dout.reshape(N * L, heads * d).to(dtype).to(DEV), st1, N, L, S, heads, d, split=False)
    torch.cuda.synchronize()
    assert chk(o1, out.detach(), L) and chk(g1[0], q.grad, L) and chk(g1[1], k.grad, S) and chk(g1[2], v.grad, S)
    assert float((state - st1).abs().max()) <= 1e-5 * float(st1.abs().max())
    assert (hip.load().cfp_linattn_ws_bytes(N, L, S, heads, d) == 0) == (N * heads >= 512 or (L <= 128 and S <= 128))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,k", [(1, 40, 50, 32, 31), (2, 33, 47, 16, 15), (1, 30, 40, 128, 7), (2, 64, 32, 8, 31)])
def test_large_depthwise_weight_gradient(B, H, W, C, k, dtype):
    x = rnd(B, C, H, W, seed=1).to(dtype).float()
    w = rnd(C, 1, k, k, seed=2, scale=0.05).requires_grad_(True)
    y = F.conv2d(x, w, None, 1, (k - 1) // 2, 1, C)
    dy = rnd(*y.shape, seed=3).to(dtype).float()
    y.backward(dy)
    dw = train_ops.dwconv_large_wgrad(nhwc(x).to(dtype).to(DEV), nhwc(dy).to(dtype).to(DEV), B, H, W, k)
    torch.cuda.synchronize()
    want = w.grad[:, 0]
    assert float((dw.cpu() - want).abs().max()) <= 3e-5 * float(want.abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W,C,k", [(2, 104, 136, 32, 31), (2, 52, 68, 64, 15), (1, 31, 33, 8, 31), (2, 26, 34, 128, 7), (1, 65, 33, 16, 7)])
def test_large_depthwise_weight_gradient_matrix_core_kernel_matches_the_vector_kernel(B, H, W, C, k, dtype):
    """dwlarge_wgrad_mfma_kernel (16-bit storage: Toeplitz view of the x rows on the matrix cores) against the VALU kernels
    (cfp_debug_set key 23) at the training step's shapes: the products are the same exact float32 values, only the summation order
    differs."""
    x = rnd(B * H * W, C, seed=11).to(dtype).to(DEV)
    dy = rnd(B * H * W, C, seed=12).to(dtype).to(DEV)
    lib = hip.load()
    try:
        lib.cfp_debug_set(23, 1)
        ref = train_ops.dwconv_large_wgrad(x, dy, B, H, W, k)
    finally:
        lib.cfp_debug_set(23, 0)
    got = train_ops.dwconv_large_wgrad(x, dy, B, H, W, k)
    again = train_ops.dwconv_large_wgrad(x, dy, B, H, W, k)
    torch.cuda.synchronize()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(got, again)


@pytest.mark.parametrize("dtype", DTYPES)
def test_batched_weight_flip_equals_the_single_launches(dtype):
    """cfp_conv2d_weight_flip_batch: every tensor of a flat parameter buffer flipped in one launch == one launch per tensor."""
    geoms = [(40, 3, 3, 16), (8, 1, 1, 264), (128, 1, 1, 128), (24, 7, 7, 8), (256, 3, 3, 392), (16, 1, 1, 8)]
    lib = hip.load()
    src_parts, rows, off_s, off_d, blocks = [], [], 0, 0, 0
    for i, (co, kh, kw, ci) in enumerate(geoms):
        n = co * kh * kw * ci
        src_parts.append(rnd(n + 24, seed=10 + i).to(dtype))                   # gaps between the tensors, as in the flat buffer
        nb = int(lib.cfp_weight_flip_blocks(n))
        rows.append([off_s + 8, off_d, co, kh, kw, ci, blocks, nb])
        off_s += n + 24; off_d += (n + 7) // 8 * 8; blocks += nb
    src = torch.cat(src_parts).to(DEV)
    dst = torch.zeros(off_d, dtype=dtype, device=DEV)
    desc = torch.tensor(rows, dtype=torch.int64).to(DEV)
    hip.call("cfp_conv2d_weight_flip_batch", src.data_ptr(), dst.data_ptr(), desc.data_ptr(), len(rows), blocks, ops.DT[dtype], hip.current_stream())
    torch.cuda.synchronize()
    for (co, kh, kw, ci), r in zip(geoms, rows):
        n = co * kh * kw * ci
        w = src[r[0]:r[0] + n].clone().view(co, kh * kw * ci)
        want = train_ops.conv2d_weight_flip(w, co, kh, kw, ci)
        ref = w.view(co, kh, kw, ci).flip(1, 2).permute(3, 1, 2, 0).reshape(ci, kh * kw * co)
        assert torch.equal(want, ref) and torch.equal(dst[r[1]:r[1] + n].view(ci, kh * kw * co), want)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [(2, 26, 34, 32, 32, 6), (1, 52, 68, 16, 24, 9), (2, 24, 36, 8, 40, 12), (1, 7, 7, 8, 8, 7)])
def test_tape_conv_with_kernel_equal_stride(B, H, W, Cin, Cout, k, dtype):
    """The GSA sub-sampling convolution (kernel = stride = window size, floor output size): its data gradient takes the
    GEMM + depth-to-space route of the tape, checked with the weight / bias gradients against torch autograd -- including the
    rows and columns past the last full patch, whose gradient is zero."""
    from cfpnet_amd.autograd_hip import P, Tape, V
    x = rnd(B, Cin, H, W, seed=1).to(dtype).float().requires_grad_(True)
    w = rnd(Cout, Cin, k, k, seed=2, scale=1.0 / math.sqrt(Cin * k * k)).to(dtype).float().requires_grad_(True)
    b = rnd(Cout, seed=3).requires_grad_(True)
    y = F.conv2d(x, w, b, stride=k)
    Ho, Wo = y.shape[2], y.shape[3]
    dy = rnd(*y.shape, seed=4).to(dtype).float()
    y.backward(dy)
    t = Tape(DEV, dtype)
    xv = V(nhwc(x.detach()).to(dtype).to(DEV))
    wp = P("w", w.detach().permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dtype).to(DEV), lambda g: g)
    bp = P("b", b.detach().to(DEV), lambda g: g)
    yv = t.conv(xv, wp, bp, B, H, W, k, k, 0, 0, Ho, Wo)
    yv.g = nhwc(dy).to(dtype).to(DEV)
    t.backward()
    torch.cuda.synchronize()
    tol = OUT_TOL[dtype] + 3e-5
    got_y = yv.t.float().cpu().reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert float((got_y - y.detach()).abs().max()) <= tol * float(y.detach().abs().max())
    got_dx = xv.g.float().cpu().reshape(B, H, W, Cin).permute(0, 3, 1, 2)
    assert float((got_dx - x.grad).abs().max()) <= tol * float(x.grad.abs().max())
    assert float(got_dx[:, :, Ho * k:, :].abs().max() if Ho * k < H else 0.0) == 0.0
    want_dw = w.grad.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin)
    assert float((wp.g.cpu() - want_dw).abs().max()) <= (3e-5 + (2e-2 if dtype != torch.float32 else 0)) * float(want_dw.abs().max())
    assert float((bp.g.cpu() - b.grad).abs().max()) <= 3e-5 * float(b.grad.abs().max()) + 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,C,R", [(16, 26 * 34, 816, 34), (16, 13 * 17, 1392, 58), (3, 50, 96, 4), (2, 17, 2048, 64), (5, 300, 384, 16)])
def test_tape_squeeze_excite_block_fused(B, HW, C, R, dtype):
    """Tape.se_block (channel sums + cfp_se_train_fwd + broadcast multiply; backward: channel dot + cfp_se_train_bwd + one multiply-add)
    against torch autograd of timm's SqueezeExcite, x * sigmoid(W2 silu(W1 mean(x) + b1) + b2), with the hidden width padded to a
    multiple of 4 the way train_model does (zero rows / columns: their gradients must come out zero)."""
    from cfpnet_amd.autograd_hip import P, Tape, V
    Rp = -(-R // 4) * 4
    x = (rnd(B, HW, C, seed=1) * 1.3 + 0.2).to(dtype).float().requires_grad_(True)
    w1 = rnd(R, C, seed=2, scale=1.0 / math.sqrt(C)).requires_grad_(True)
    b1 = rnd(R, seed=3, scale=0.3).requires_grad_(True)
    w2 = rnd(C, R, seed=4, scale=1.0 / math.sqrt(R)).requires_grad_(True)
    b2 = rnd(C, seed=5, scale=0.3).requires_grad_(True)
    gate = torch.sigmoid(F.silu(x.mean(1) @ w1.t() + b1) @ w2.t() + b2)
    y = x * gate[:, None, :]
    dy = rnd(B, HW, C, seed=6).to(dtype).float()
    y.backward(dy)

    t = Tape(DEV, dtype)
    w1p = torch.zeros(Rp, C); w1p[:R] = w1.detach()
    b1p = torch.zeros(Rp); b1p[:R] = b1.detach()
    w2p = torch.zeros(C, Rp); w2p[:, :R] = w2.detach()
    ps = [P("w1", w1p.to(DEV), lambda g: g), P("b1", b1p.to(DEV), lambda g: g), P("w2", w2p.to(DEV), lambda g: g), P("b2", b2.detach().to(DEV), lambda g: g)]
    xv = V(x.detach().reshape(B * HW, C).to(dtype).to(DEV))
    yv = t.se_block(xv, *ps, B, HW)
    yv.g = dy.reshape(B * HW, C).to(dtype).to(DEV)
    t.backward()
    torch.cuda.synchronize()
    tol = OUT_TOL[dtype]
    assert float((yv.t.float().cpu().reshape(B, HW, C) - y.detach()).abs().max()) <= tol * float(y.detach().abs().max())
    assert float((xv.g.float().cpu().reshape(B, HW, C) - x.grad).abs().max()) <= (tol + 2e-5) * float(x.grad.abs().max())
    # parameter gradients: float32 sums of products of 16-bit-rounded inputs -> tight bounds in every mode
    pt = 2e-4 if dtype == torch.float32 else 2e-2
    for got, want in ((ps[0].g.cpu()[:R], w1.grad), (ps[1].g.cpu()[:R], b1.grad), (ps[2].g.cpu()[:, :R], w2.grad), (ps[3].g.cpu(), b2.grad)):
        assert float((got - want).abs().max()) <= pt * float(want.abs().max()) + 1e-6
    if Rp > R:
        assert float(ps[0].g[R:].abs().max()) == 0 and float(ps[1].g[R:].abs().max()) == 0 and float(ps[2].g[:, R:].abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,Ca,Cb", [(1000, 64, 40), (37, 256, 232), (5000, 8, 128)])
def test_tape_concat_and_split(rows, Ca, Cb, dtype):
    """Tape.concat: [a | b] in one launch (cfp_copy_rows2), its backward the two slices in one launch; exact copies."""
    from cfpnet_amd.autograd_hip import Tape, V
    a, b = rnd(rows, Ca, seed=1).to(dtype).to(DEV), rnd(rows, Cb, seed=2).to(dtype).to(DEV)
    for need_b in (True, False):
        t = Tape(DEV, dtype)
        av, bv = V(a), V(b, needs_grad=need_b)
        y = t.concat(av, bv)
        g = rnd(rows, Ca + Cb, seed=3).to(dtype).to(DEV)
        y.g = g
        t.backward()
        torch.cuda.synchronize()
        assert torch.equal(y.t, torch.cat([a, b], 1))
        assert torch.equal(av.g, g[:, :Ca]) and (bv.g is None if not need_b else torch.equal(bv.g, g[:, Ca:]))


# launches with >= 1e5 pixel rows take the 128-wide weight-gradient tiles (conv_bwd.hip wgrad_plan16): one case per tile shape,
# ragged Cout / K, the im2col path with padding and the pointwise path
BIG_WGRAD = [(2, 224, 232, 32, 128, 3, 1, (1, 1, 1, 1)),      # 128 x 128 tiles, K = 288
             (1, 320, 328, 80, 32, 3, 1, (1, 1, 1, 1)),       # 64 x 128 (Cout = 32: half a Cout tile), K = 720
             (1, 330, 310, 128, 200, 1, 1, (0, 0, 0, 0)),     # pointwise, 128 x 128, ragged Cout
             (1, 330, 310, 64, 136, 1, 1, (0, 0, 0, 0)),      # 128 x 64 (K = 64)
             (2, 230, 226, 24, 72, 3, 2, (0, 0, 1, 1))]       # stride 2: 25 k rows only -> the 64 x 64 tile; control


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", BIG_WGRAD)
def test_conv_weight_gradient_many_rows(case, dtype):
    B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
    x, w, dy, dx_ref, dw_ref, Ho, Wo = _ref(case, dtype)
    xd, dyd = nhwc(x).to(dtype).to(DEV), nhwc(dy).to(dtype).to(DEV)
    dw = train_ops.conv2d_wgrad(xd, dyd, B, H, W, k, k, s, pt, pl, Ho, Wo)
    dw3 = train_ops.conv2d_wgrad(xd, dyd, B, H, W, k, k, s, pt, pl, Ho, Wo)
    torch.cuda.synchronize()
    got = dw.cpu().reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    err = float((got - dw_ref).abs().max()) / float(dw_ref.abs().max())
    assert err < 1e-4, (case, dtype, err)                      # f32 sums of ~1e5 exact products
    assert torch.equal(dw3, dw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,k", [(32, 31), (64, 15), (128, 7), (16, 31)])
def test_toeplitz_band_table_on_device(C, k, dtype):
    """cfp_dwconv_large_toeplitz (the band table of the matrix-core large-kernel depthwise conv, rebuilt from the float32 master weights
    every training step) == the same gather written with torch, for the kernel and for its 180-degree rotation (data gradient)."""
    w = rnd(C, k, k, seed=1)
    halo = (k - 1) // 2
    lm = (halo + 7) // 8 * 8
    nh = (16 + lm + halo + 31) // 32
    lane, h, e = torch.arange(64), torch.arange(nh), torch.arange(8)
    kx = 32 * h[:, None, None] + 8 * (lane[None, :, None] // 16) + e[None, None, :] - (lm - halo) - (lane[None, :, None] % 16)
    ok = (kx >= 0) & (kx < k)
    for flip in (False, True):
        src = w.flip(1, 2) if flip else w
        want = torch.where(ok[None, None], src[:, :, kx.clamp(0, k - 1)], torch.zeros(())).to(dtype).reshape(-1)
        got = ops.toeplitz_bands_dev(w.to(DEV), dtype, flip=flip)
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES[:6] + BIG_WGRAD[:4])
def test_conv_weight_gradient_with_fused_bias_gradient(case, dtype):
    """cfp_conv2d_wgrad_bias: db = column sums of dY from the weight-gradient launch itself (a fragment of ones on the matrix cores),
    with its own beta; dW unchanged (bit for bit) by the extra product."""
    B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
    x, w, dy, dx_ref, dw_ref, Ho, Wo = _ref(case, dtype)
    xd, dyd = nhwc(x).to(dtype).to(DEV), nhwc(dy).to(dtype).to(DEV)
    dw0 = train_ops.conv2d_wgrad(xd, dyd, B, H, W, k, k, s, pt, pl, Ho, Wo)
    db = torch.full((Cout + 3,), 2.0, device=DEV)
    dw1 = train_ops.conv2d_wgrad(xd, dyd, B, H, W, k, k, s, pt, pl, Ho, Wo, db=db, beta_b=0.5)
    torch.cuda.synchronize()
    assert torch.equal(dw1, dw0)
    want = nhwc(dy).to(dtype).float().sum(0)
    got = db.cpu()
    assert float((got[:Cout] - (1.0 + want)).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-4
    assert float((got[Cout:] - 2.0).abs().max()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_batched_weight_gradient_reduction_is_bit_identical(dtype):
    """cfp_conv2d_wgrad_deferred + cfp_wgrad_reduce_jobs (many layers' slab reductions in one launch, the job table in the kernel
    arguments) == the per-layer launches, bit for bit: with and without the fused bias gradient, with beta, with more jobs than one
    launch holds (48) and with two jobs that write the same tensor (they must land in separate launches, in order)."""
    cases = (CASES[:6] + BIG_WGRAD) * 6                      # 66 calls (a few finish in place): two launches
    q = train_ops.WgradQueue()
    want, got = [], []
    for n, case in enumerate(cases):
        B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
        Ho, Wo = (H + pt + pb - k) // s + 1, (W + pl + pr - k) // s + 1
        xd = rnd(B * H * W, Cin, seed=3 * n).to(dtype).to(DEV)
        dyd = rnd(B * Ho * Wo, Cout, seed=3 * n + 1).to(dtype).to(DEV)
        with_bias = dtype != torch.float32 and n % 2 == 1
        beta = 0.5 if n % 3 == 2 else 0.0
        base = rnd(Cout, k * k * Cin, seed=3 * n + 2).to(DEV)
        db0 = torch.full((Cout,), 2.0, device=DEV) if with_bias else None
        a, adb = base.clone(), (db0.clone() if with_bias else None)
        b, bdb = base.clone(), (db0.clone() if with_bias else None)
        train_ops.conv2d_wgrad(xd, dyd, B, H, W, k, k, s, pt, pl, Ho, Wo, dw=a, beta=beta, db=adb, beta_b=0.25)
        train_ops.conv2d_wgrad(xd, dyd, B, H, W, k, k, s, pt, pl, Ho, Wo, dw=b, beta=beta, db=bdb, beta_b=0.25, queue=q)
        want.append((a, adb)); got.append((b, bdb))
    # the same tensor twice: dw = grad, then dw = 1 * dw + grad
    B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = BIG_WGRAD[0]
    Ho, Wo = (H + pt + pb - k) // s + 1, (W + pl + pr - k) // s + 1
    xd, dyd = rnd(B * H * W, Cin, seed=900).to(dtype).to(DEV), rnd(B * Ho * Wo, Cout, seed=901).to(dtype).to(DEV)
    twice_ref = torch.empty(Cout, k * k * Cin, device=DEV)
    train_ops.conv2d_wgrad(xd, dyd, B, H, W, k, k, s, pt, pl, Ho, Wo, dw=twice_ref, beta=0.0)
    train_ops.conv2d_wgrad(xd, dyd, B, H, W, k, k, s, pt, pl, Ho, Wo, dw=twice_ref, beta=1.0)
    twice = torch.empty(Cout, k * k * Cin, device=DEV)
    train_ops.conv2d_wgrad(xd, dyd, B, H, W, k, k, s, pt, pl, Ho, Wo, dw=twice, beta=0.0, queue=q)
    train_ops.conv2d_wgrad(xd, dyd, B, H, W, k, k, s, pt, pl, Ho, Wo, dw=twice, beta=1.0, queue=q)
    assert len(q.jobs) > 48
    q.flush()
    assert not q.jobs and not q.keep
    torch.cuda.synchronize()
    for n, ((a, adb), (b, bdb)) in enumerate(zip(want, got)):
        assert torch.equal(a, b), (n, cases[n])
        if adb is not None:
            assert torch.equal(adb, bdb), (n, cases[n])
    assert torch.equal(twice, twice_ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_and_depthwise_parameter_gradients_through_the_batched_reduction(dtype):
    """cfp_layernorm_bwd_deferred / cfp_dwconv3x3_wgrad_deferred leave their finishing sums to cfp_wgrad_reduce_jobs (the job table of the
    dense weight gradients): same values as the immediate calls up to float32 summation order, dx untouched by the deferral."""
    q = train_ops.WgradQueue()
    checks = []
    for n, (rows, C) in enumerate([(9600, 128), (38400, 64), (153600, 32), (240, 128), (77, 64)]):
        x, dy = rnd(rows, C, seed=20 + n).to(dtype).to(DEV), rnd(rows, C, seed=30 + n).to(dtype).to(DEV)
        g = (1.0 + 0.1 * rnd(C, seed=40 + n)).to(DEV)
        dx0, dg0, db0 = train_ops.layernorm_bwd(x, dy, g, 1e-5)
        dx1, dg1, db1 = train_ops.layernorm_bwd(x, dy, g, 1e-5, queue=q)
        checks.append((dx0, dx1, True)); checks.append((dg0, dg1, False)); checks.append((db0, db1, False))
    for n, (B, H, W, C, s) in enumerate([(2, 26, 34, 816, 1), (2, 52, 68, 224, 2), (1, 13, 17, 1392, 1)]):
        pt, pl = (1, 1) if s == 1 else (0, 0)
        Ho, Wo = (H + 2 - 3) // s + 1 if s == 1 else -(-H // s), (W + 2 - 3) // s + 1 if s == 1 else -(-W // s)
        x, dy = rnd(B * H * W, C, seed=50 + n).to(dtype).to(DEV), rnd(B * Ho * Wo, C, seed=60 + n).to(dtype).to(DEV)
        base = rnd(9, C, seed=70 + n).to(DEV)
        beta = 0.5 if n == 1 else 0.0
        a, b = base.clone(), base.clone()
        train_ops.dwconv3x3_wgrad(x, dy, B, H, W, s, pt, pl, Ho, Wo, dw=a, beta=beta)
        train_ops.dwconv3x3_wgrad(x, dy, B, H, W, s, pt, pl, Ho, Wo, dw=b, beta=beta, queue=q)
        checks.append((a, b, False))
    assert len(q.jobs) == 8
    q.flush()
    torch.cuda.synchronize()
    for want, got, exact in checks:
        if exact:
            assert torch.equal(want, got)
        else:
            assert float((want - got).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6
