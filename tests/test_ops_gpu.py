"""Kernel-level numerics: every HIP entry point against a plain PyTorch fp32 reference of the
same op (CPU), in f32 (tight) and bf16 (storage-rounding) tolerances.  Runs on the GPU box."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cfpnet_amd import hip, ops  # noqa: E402

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
HALF = [torch.bfloat16, torch.float16]      # 16-bit storage: the MFMA fast paths


def tol(dtype):
    return {torch.float32: (2e-5, 2e-5), torch.bfloat16: (2.5e-2, 2.5e-2), torch.float16: (4e-3, 4e-3)}[dtype]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def q(x, dtype):
    """Quantise a reference input the way the device tensor is stored."""
    return x.to(dtype).float()


def to_act(x2d: torch.Tensor, dtype, ld=None, c0=0) -> ops.Act:
    rows, C = x2d.shape
    ld = ld or C
    buf = torch.zeros(rows, ld, dtype=dtype, device=DEV)
    buf[:, c0:c0 + C] = x2d.to(dtype).to(DEV)
    return ops.Act(buf, c0, C)


def nhwc(x):   # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def from_nhwc(t2d, B, H, W):
    return t2d.float().cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2)


def close(got, ref, dtype, what=""):
    rt, at = tol(dtype)
    err = (got - ref).abs()
    bound = at * ref.abs().max().clamp(min=1e-3) + rt * ref.abs()
    assert bool((err <= bound).all()), f"{what}: max err {float(err.max()):.3e} ref max {float(ref.abs().max()):.3e}"


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pads(t,l,b,r)
    (2, 9, 11, 8, 40, 3, 2, (0, 0, 1, 1)),        # stem-like, SAME asymmetric (odd sizes)
    (1, 12, 16, 40, 16, 3, 1, (1, 1, 1, 1)),      # Cout 16 tile
    (2, 10, 14, 16, 64, 3, 2, (0, 0, 1, 1)),      # edge-residual expand, even input
    (1, 15, 20, 392, 256, 3, 1, (1, 1, 1, 1)),    # up1 first conv, K = 3528
    (2, 8, 8, 136, 816, 1, 1, (0, 0, 0, 0)),      # pointwise, Cout tail (816 = 6*128 + 48)
    (1, 30, 40, 128, 128, 6, 6, (0, 0, 0, 0)),    # GSA sr conv: kernel = stride = 6
    (1, 13, 17, 32, 32, 3, 1, (1, 1, 1, 1)),      # Cout 32 tile, ragged M
    (1, 1, 300, 256, 256, 1, 1, (0, 0, 0, 0)),    # Linear (rows = 300)
    (1, 1, 70, 8, 32, 1, 1, (0, 0, 0, 0)),        # K = 8 (ToF first layer)
    (2, 60, 80, 64, 64, 9, 9, (0, 0, 0, 0)),      # GSA sr conv at 1/8: M = 96, K = 5184 -> split-K
    (2, 15, 20, 1392, 232, 1, 1, (0, 0, 0, 0)),   # 1/32-scale project conv: M = 600, K = 1392 -> split-K
    (16, 1, 1, 1056, 40, 1, 1, (0, 0, 0, 0)),     # squeeze-excite reduce FC on [B, C] vectors: f32 takes the few-row kernel (M <= 64)
    (1, 1, 17, 136, 528, 1, 1, (0, 0, 0, 0)),     # ... expand FC, 17 rows = two row chunks
    (4, 1, 16, 2064, 88, 1, 1, (0, 0, 0, 0)),     # ... M = 64, the last few-row size
    (1, 1, 1, 48, 16, 1, 1, (0, 0, 0, 0)),        # ... a single row
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case, dtype):
    B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
    x = q(rnd(B, Cin, H, W, seed=1), dtype)
    w = q(rnd(Cout, Cin, k, k, seed=2, scale=1.0 / math.sqrt(Cin * k * k)), dtype)
    scale = rnd(Cout, seed=3).abs() + 0.5
    shift = rnd(Cout, seed=4)
    Ho = (H + pt + pb - k) // s + 1
    Wo = (W + pl + pr - k) // s + 1
    res = q(rnd(B, Cout, Ho, Wo, seed=5), dtype)
    ref = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, None, s)
    ref = F.silu(ref * scale[None, :, None, None] + shift[None, :, None, None]) + res
    xa = to_act(nhwc(x), dtype, ld=Cin + 16, c0=8)
    wa = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dtype).to(DEV)
    ra = to_act(nhwc(res), dtype)
    out = ops.new_act(B * Ho * Wo, Cout, dtype, DEV, ld=Cout + 24, zero=True).slice(0, Cout)
    out = ops.Act(out.buf, 16, Cout)
    nws = ops.conv2d_ws_bytes(B * Ho * Wo, Cout, k * k * Cin, ops.DT[dtype])
    ws = torch.empty(max(nws // 4, 1), device=DEV) if nws else None
    ops.conv2d(xa, wa, scale.to(DEV), shift.to(DEV), out, B, H, W, k, k, s, pt, pl, Ho, Wo, hip.ACT_SILU, ra, ws)
    torch.cuda.synchronize()
    close(from_nhwc(out.torch(), B, Ho, Wo), ref, dtype, f"conv {case} (splitk ws {nws})")
    if nws:   # the un-split kernel must agree
        out.buf.zero_()
        ops.conv2d(xa, wa, scale.to(DEV), shift.to(DEV), out, B, H, W, k, k, s, pt, pl, Ho, Wo, hip.ACT_SILU, ra, None)
        close(from_nhwc(out.torch(), B, Ho, Wo), ref, dtype, f"conv {case} (no ws)")
    # the slice neighbours must be untouched
    assert float(out.buf[:, :16].abs().max()) == 0 and float(out.buf[:, 16 + Cout:].abs().max()) == 0


def _conv_ref_and_args(case, dtype, seed=1):
    B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
    x = q(rnd(B, Cin, H, W, seed=seed), dtype)
    w = q(rnd(Cout, Cin, k, k, seed=seed + 1, scale=1.0 / math.sqrt(Cin * k * k)), dtype)
    scale = rnd(Cout, seed=seed + 2).abs() + 0.5
    shift = rnd(Cout, seed=seed + 3)
    Ho = (H + pt + pb - k) // s + 1
    Wo = (W + pl + pr - k) // s + 1
    res = q(rnd(B, Cout, Ho, Wo, seed=seed + 4), dtype)
    ref = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, None, s)
    ref = F.silu(ref * scale[None, :, None, None] + shift[None, :, None, None]) + res
    xa = to_act(nhwc(x), dtype, ld=Cin + 16, c0=8)
    wa = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dtype).to(DEV)
    ra = to_act(nhwc(res), dtype)
    return ref, xa, wa, scale.to(DEV), shift.to(DEV), ra, Ho, Wo


GEN2_VARIANTS = list(range(22))
GEN2_KGROUPS = (19, 20, 21)          # eight-wave variants (two K groups per workgroup): the in-workgroup alternative to split-K


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("variant", GEN2_VARIANTS)
def test_conv2d_gen2_every_tile_variant(variant, dtype):
    """Every second-generation (LDS-DMA staged) tile configuration on every conv case, forced through
    the debug knob, un-split and with 4 K-splits."""
    lib = hip.load()
    try:
        lib.cfp_debug_set(0, variant)
        for case in CONV_CASES:
            B, H, W, Cin, Cout, k, s, _ = case
            ref, xa, wa, scale, shift, ra, Ho, Wo = _conv_ref_and_args(case, dtype)
            for splits in ((1,) if variant in GEN2_KGROUPS else (1, 4)):
                lib.cfp_debug_set(1, splits)
                out = ops.new_act(B * Ho * Wo, Cout, dtype, DEV, ld=Cout + 24, zero=True)
                out = ops.Act(out.buf, 16, Cout)
                ws = torch.empty(splits * B * Ho * Wo * Cout, device=DEV)
                ops.conv2d(xa, wa, scale, shift, out, B, H, W, k, k, s, case[7][0], case[7][1], Ho, Wo, hip.ACT_SILU, ra, ws)
                torch.cuda.synchronize()
                close(from_nhwc(out.torch(), B, Ho, Wo), ref, dtype, f"gen2 v{variant} splits {splits} conv {case}")
                assert float(out.buf[:, :16].abs().max()) == 0 and float(out.buf[:, 16 + Cout:].abs().max()) == 0
    finally:
        lib.cfp_debug_set(0, -1)
        lib.cfp_debug_set(1, -1)


# ---- float32 storage with split-precision (f16x3) matrix math: csrc/conv_igemm_x3.hip ------------------------------------------
X3_TOL = 4e-6       # relative to the largest |reference| entry; the float32 MFMA kernel measures ~1e-6 on the same problems


def _x3_problem(case, seed=1, xscale=1.0):
    """float64 reference + device operands of a conv case for the f16x3 kernels (weights pre-split by cfp_pack_w_x3)."""
    B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
    x = rnd(B, Cin, H, W, seed=seed) * xscale
    w = rnd(Cout, Cin, k, k, seed=seed + 1, scale=1.0 / math.sqrt(Cin * k * k))
    scale = rnd(Cout, seed=seed + 2).abs() + 0.5
    shift = rnd(Cout, seed=seed + 3) * xscale
    Ho = (H + pt + pb - k) // s + 1
    Wo = (W + pl + pr - k) // s + 1
    res = rnd(B, Cout, Ho, Wo, seed=seed + 4) * xscale
    ref = F.conv2d(F.pad(x.double(), (pl, pr, pt, pb)), w.double(), None, s)
    ref = F.silu(ref * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]) + res.double()
    xa = to_act(nhwc(x), torch.float32, ld=Cin + 16, c0=8)
    w32 = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(DEV)
    ra = to_act(nhwc(res), torch.float32)
    return ref, xa, w32, ops.pack_w_x3(w32), scale.to(DEV), shift.to(DEV), ra, Ho, Wo


def _x3_close(got, ref, what, tol=X3_TOL):
    err = float((got.double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
    assert err <= tol, f"{what}: max err / max |ref| = {err:.3e}"
    return err


def test_pack_w_x3_layout():
    """hi + lo of the packed operand reproduce the float32 weights to 2^-21, in the documented lane order, zero padded."""
    co, K = 5, 72
    w = (rnd(co, K, seed=3) * 0.1).to(DEV)
    pk = ops.pack_w_x3(w).float().cpu().reshape(co, -1, 2, 32)
    assert pk.shape[1] == 3
    r = torch.arange(32)
    pos = ((r % 16) // 4) * 8 + r % 4 + 4 * (r // 16)
    back = torch.zeros(co, 96)
    for ks in range(3):
        back[:, ks * 32 + r] = pk[:, ks, 0, pos] + pk[:, ks, 1, pos]
    assert float((back[:, :K] - w.cpu()).abs().max()) <= 2.0 ** -21 * float(w.abs().max())
    assert float(back[:, K:].abs().max()) == 0


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_x3(case):
    """Every conv case through the automatic f16x3 plan (with and without its split-K workspace) against a float64 reference:
    as tight as the float32 MFMA kernel, three decimal orders inside the 1e-3 gate."""
    B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
    ref, xa, w32, wx, scale, shift, ra, Ho, Wo = _x3_problem(case)
    nws = ops.conv2d_ws_bytes(B * Ho * Wo, Cout, k * k * Cin, hip.F32X3)
    for ws in ([torch.empty(max(nws // 4, 1), device=DEV), None] if nws else [None]):
        out = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV, ld=Cout + 24, zero=True)
        out = ops.Act(out.buf, 16, Cout)
        ops.conv2d(xa, wx, scale, shift, out, B, H, W, k, k, s, pt, pl, Ho, Wo, hip.ACT_SILU, ra, ws)
        torch.cuda.synchronize()
        _x3_close(from_nhwc(out.torch(), B, Ho, Wo), ref, f"x3 conv {case} (ws {nws if ws is not None else 0})")
        assert float(out.buf[:, :16].abs().max()) == 0 and float(out.buf[:, 16 + Cout:].abs().max()) == 0


@pytest.mark.parametrize("variant", list(range(37)))      # 0-21: conv_igemm2.hip's ids, 22-27: f16x3 only, 28-33: f16x3 A-direct, 34-36: occupancy builds
def test_conv2d_x3_every_tile_variant(variant):
    """Every f16x3 tile configuration on every conv case, forced through the debug knob (400 + v), un-split and with 4 K-splits."""
    lib = hip.load()
    try:
        lib.cfp_debug_set(0, 400 + variant)
        for case in CONV_CASES:
            B, H, W, Cin, Cout, k, s, pads = case
            ref, xa, w32, wx, scale, shift, ra, Ho, Wo = _x3_problem(case)
            for splits in ((1,) if (variant in GEN2_KGROUPS or variant in ops.X3_ADIRECT or variant in ops.X3_OCC) else (1, 4)):
                lib.cfp_debug_set(1, splits)
                out = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV, ld=Cout + 24, zero=True)
                out = ops.Act(out.buf, 16, Cout)
                ws = torch.empty(splits * B * Ho * Wo * Cout, device=DEV)
                ops.conv2d(xa, wx, scale, shift, out, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, ws)
                torch.cuda.synchronize()
                _x3_close(from_nhwc(out.torch(), B, Ho, Wo), ref, f"x3 v{variant} splits {splits} conv {case}")
                assert float(out.buf[:, :16].abs().max()) == 0 and float(out.buf[:, 16 + Cout:].abs().max()) == 0
    finally:
        lib.cfp_debug_set(0, -1)
        lib.cfp_debug_set(1, -1)


@pytest.mark.parametrize("variant", [4, 13, 14, 12, 17, 0])
def test_conv2d_x3_ticketed_split_k_is_bit_identical_to_the_reduce_launch(variant):
    """CFP_CONV_WS_TICKETS: the last workgroup at an output tile sums the slabs (in split order, from memory) and runs the epilogue -- EQUAL to
    the two-launch form (slabs + splitk_reduce_kernel) for 2 .. 8 splits, ragged tiles, residual + activation; the ticket area comes back zeroed,
    also after several launches on one workspace (a captured graph replays it)."""
    lib = hip.load()
    cases = [(1, 15, 20, 1824, 304, 1, 1, (0, 0, 0, 0)), (2, 8, 8, 136, 816, 1, 1, (0, 0, 0, 0)), (1, 9, 11, 72, 40, 3, 1, (1, 1, 1, 1)),
             (1, 30, 40, 960, 176, 1, 1, (0, 0, 0, 0)), (3, 7, 9, 96, 64, 3, 1, (1, 1, 1, 1)), (1, 1, 515, 128, 128, 1, 1, (0, 0, 0, 0))]
    try:
        lib.cfp_debug_set(0, 400 + variant)
        for case in cases:
            B, H, W, Cin, Cout, k, s, pads = case
            ref, xa, w32, wx, scale, shift, ra, Ho, Wo = _x3_problem(case)
            for splits in (2, 3, 8):
                lib.cfp_debug_set(1, splits)
                slab_bytes = splits * B * Ho * Wo * Cout * 4
                plain = torch.empty(slab_bytes // 4, device=DEV)
                tws = ops.ticket_ws(slab_bytes, DEV)
                x0 = xa.buf.clone()
                for it in range(3):      # the same workspace with DIFFERENT data every time: a slab line left in some L2 by the launch before would show
                    xa.buf.copy_(x0 * (1.0 + 0.37 * it))
                    got = []
                    for ws in (plain, tws):
                        out = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV, ld=Cout + 24, zero=True)
                        out = ops.Act(out.buf, 16, Cout)
                        ops.conv2d(xa, wx, scale, shift, out, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, ws)
                        torch.cuda.synchronize()
                        got.append(out.buf.clone())
                    assert int(tws[: hip.CONV_TICKET_BYTES // 4].view(torch.int32).abs().max()) == 0, (variant, case, splits)
                    assert torch.equal(got[0], got[1]), (variant, case, splits, it)
                    if it == 0:
                        _x3_close(from_nhwc(ops.Act(got[1], 16, Cout).torch(), B, Ho, Wo), ref, f"x3 ticketed v{variant} splits {splits} {case}")
                xa.buf.copy_(x0)
        # per-image weights (the project GEMM of a squeeze-excite block on single images: the layer the engine splits)
        lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1)
        B, hw, K, Cout = 2, 300, 1824, 304
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B * hw, K, generator=g).to(DEV)
        x0 = x.clone()
        w = (torch.randn(B, Cout, K, generator=g) / K ** 0.5).to(DEV)
        wx = torch.stack([ops.pack_w_x3(w[b]) for b in range(B)]).contiguous()
        sc, sh = torch.rand(Cout, generator=g).to(DEV) + 0.5, torch.randn(Cout, generator=g).to(DEV)
        _, sp = ops.conv2d_plan(B * hw, Cout, K, hip.F32X3, hw, B, 1, 1)
        assert sp > 1, "the plan is expected to split this per-image GEMM"
        xa = ops.Act(x, 0, K)
        got = []
        for ws in (torch.empty(sp * B * hw * Cout, device=DEV), ops.ticket_ws(sp * B * hw * Cout * 4, DEV)):
            out = ops.new_act(B * hw, Cout, torch.float32, DEV)
            for it in range(3):      # back to back on one stream, the last launch on the original data
                x.copy_(x0 * (2.0 - it if it < 2 else 1.0))
                ops.conv2d(xa, wx, sc, sh, out, B, hw, 1, 1, 1, 1, 0, 0, hw, 1, hip.ACT_NONE, None, ws, per_image_weights=True)
            torch.cuda.synchronize()
            got.append(out.torch().clone())
        assert torch.equal(got[0], got[1])
        want = torch.stack([x[b * hw:(b + 1) * hw].double() @ w[b].double().t() for b in range(B)]).reshape(B * hw, Cout) * sc.double() + sh.double()
        assert float((got[1].double() - want).abs().max() / want.abs().max()) < 2e-5
    finally:
        lib.cfp_debug_set(0, -1)
        lib.cfp_debug_set(1, -1)


@pytest.mark.parametrize("pair", [(28, 26), (29, 14), (30, 13), (31, 23), (32, 16), (34, 13), (35, 16), (36, 15)])
def test_conv2d_x3_a_direct_loop_is_bit_identical_to_the_staged_loop(pair):
    """The A-direct K loop (A values global -> registers, only the W tile through LDS) against the staged loop of the same tile -- and the
    occupancy builds (ids 34-36: the same source under a register budget) against theirs: same
    products in the same order, so the float32 results must be EQUAL -- K lengths of 1, 2, 3, 4 and many K-steps, 1x1 and 3x3 (taps that
    straddle K-steps: Cin = 8, 40, 392), stride 2, ragged row and channel tails."""
    ad, staged = pair
    lib = hip.load()
    cases = [(1, 1, 300, 32, 64, 1, 1, (0, 0, 0, 0)), (1, 1, 700, 40, 136, 1, 1, (0, 0, 0, 0)), (1, 9, 11, 8, 40, 3, 1, (1, 1, 1, 1)),
             (2, 8, 8, 136, 816, 1, 1, (0, 0, 0, 0)), (1, 15, 20, 392, 256, 3, 1, (1, 1, 1, 1)), (2, 15, 20, 1392, 232, 1, 1, (0, 0, 0, 0)),
             (1, 1, 77, 64, 16, 1, 1, (0, 0, 0, 0)), (2, 17, 23, 40, 72, 3, 2, (0, 1, 0, 1)), (1, 1, 515, 128, 128, 1, 1, (0, 0, 0, 0)),
             (3, 7, 9, 96, 64, 3, 1, (1, 1, 1, 1))]
    try:
        for case in cases:
            B, H, W, Cin, Cout, k, s, pads = case
            ref, xa, w32, wx, scale, shift, ra, Ho, Wo = _x3_problem(case)
            got = []
            for v in (ad, staged):
                lib.cfp_debug_set(0, 400 + v)
                lib.cfp_debug_set(1, 1)
                out = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV)
                ops.conv2d(xa, wx, scale, shift, out, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
                torch.cuda.synchronize()
                got.append(out.torch().clone())
            assert torch.equal(got[0], got[1]), (pair, case)
            _x3_close(from_nhwc(got[0], B, Ho, Wo), ref, f"x3 a-direct v{ad} {case}")
    finally:
        lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1)


@pytest.mark.parametrize("variant", [13, 14, 26, 1, 15, 16, 2, 22])
def test_conv2d_x3_fragment_pipelined_loop_is_bit_identical_to_the_plain_loop(variant):
    """The K loop that reads the next step's fragments while the current step's MFMAs run (default) against the plain loop
    (cfp_debug_set(28, 1)): same products in the same order, so the float32 results must be EQUAL -- on K lengths of 1, 2, 3 and many
    K-steps, with and without split-K."""
    lib = hip.load()
    cases = [(1, 1, 300, 32, 64, 1, 1, (0, 0, 0, 0)), (1, 1, 700, 40, 136, 1, 1, (0, 0, 0, 0)), (1, 9, 11, 8, 40, 3, 1, (1, 1, 1, 1)),
             (2, 8, 8, 136, 816, 1, 1, (0, 0, 0, 0)), (1, 15, 20, 392, 256, 3, 1, (1, 1, 1, 1)), (2, 15, 20, 1392, 232, 1, 1, (0, 0, 0, 0))]
    try:
        lib.cfp_debug_set(0, 400 + variant)
        for case in cases:
            B, H, W, Cin, Cout, k, s, pads = case
            ref, xa, w32, wx, scale, shift, ra, Ho, Wo = _x3_problem(case)
            for splits in (1, 3):
                lib.cfp_debug_set(1, splits)
                got = []
                for plain in (0, 1):
                    lib.cfp_debug_set(28, plain)
                    out = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV)
                    ws = torch.empty(splits * B * Ho * Wo * Cout, device=DEV)
                    ops.conv2d(xa, wx, scale, shift, out, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, ws)
                    torch.cuda.synchronize()
                    got.append(out.torch().clone())
                assert torch.equal(got[0], got[1]), (variant, case, splits)
                _x3_close(from_nhwc(got[0], B, Ho, Wo), ref, f"x3 pipelined v{variant} {case}")
    finally:
        lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1); lib.cfp_debug_set(28, 0)


@pytest.mark.parametrize("xscale", [1e-2, 1e-4, 3e3])
def test_conv2d_x3_small_and_large_magnitudes(xscale):
    """Activations far from 1: at 1e-2 the lo halves are IEEE-half SUBNORMALS (the matrix core must take them as they are), at 1e-4 the
    hi halves are; at 3e3 sums approach the half range.  The error stays relative to the data, not to 1."""
    case = (2, 15, 20, 136, 232, 1, 1, (0, 0, 0, 0))
    B, H, W, Cin, Cout, k, s, pads = case
    ref, xa, w32, wx, scale, shift, ra, Ho, Wo = _x3_problem(case, xscale=xscale)
    out = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV)
    ops.conv2d(xa, wx, scale, shift, out, B, H, W, k, k, s, 0, 0, Ho, Wo, hip.ACT_SILU, ra, None)
    torch.cuda.synchronize()
    # below ~6e-5 a half has fewer than 11 bits: the absolute error floor is 2^-25 per term (harmless next to O(1) activations)
    _x3_close(from_nhwc(out.torch(), B, Ho, Wo), ref, f"x3 magnitudes {xscale}", tol=4e-6 if xscale >= 1e-2 else 2e-4)


def test_conv2d_x3_dynamic_range_limit_is_what_the_docs_say():
    """the f16x3 split holds a float32 value as TWO IEEE halves, so its range is 2 x 65504: up to 6.5e4 with ~22 bits, from
    there to 1.31e5 with the 11 bits of the lo half alone (hi saturates at 65504), beyond that lo overflows (non-finite result).  The
    float32 reference has no such limit; `Deltar(check_finite=True)` / `dtype=torch.float32` are the documented ways out.  Activations
    of a BatchNorm network are O(1) ... O(100)."""
    case = (2, 15, 20, 136, 232, 1, 1, (0, 0, 0, 0))
    B, H, W, Cin, Cout, k, s, pads = case
    for xscale, tol, finite in ((1.4e4, 4e-6, True), (2.7e4, 2e-3, True), (1e5, None, False)):      # max |x| ~ 4.5 sigma: 6.3e4 / 1.2e5 / 4.5e5
        ref, xa, w32, wx, scale, shift, ra, Ho, Wo = _x3_problem(case, xscale=xscale)
        out = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV)
        ops.conv2d(xa, wx, scale, shift, out, B, H, W, k, k, s, 0, 0, Ho, Wo, hip.ACT_NONE, None, None)
        torch.cuda.synchronize()
        got = from_nhwc(out.torch(), B, Ho, Wo)
        assert bool(torch.isfinite(got).all()) == finite, xscale
        if finite:
            x = rnd(B, Cin, H, W, seed=1) * xscale                        # _x3_problem's own draws (seed 1: x, seed 2: w)
            w = rnd(Cout, Cin, k, k, seed=2, scale=1.0 / math.sqrt(Cin * k * k))
            y = F.conv2d(x.double(), w.double()) * scale.cpu().double()[None, :, None, None] + shift.cpu().double()[None, :, None, None]
            _x3_close(got.cpu(), y, f"x3 range {xscale}", tol=tol)


@pytest.mark.parametrize("case", [(2, 15, 20, 30, 40, 64, 16, 32), (1, 8, 10, 16, 20, 128, 40, 64), (2, 5, 7, 10, 14, 256, 56, 128),
                                  (1, 4, 5, 8, 10, 256, 136, 256), (1, 7, 9, 13, 21, 32, 8, 16), (1, 3, 3, 9, 33, 64, 4, 96), (3, 6, 6, 12, 12, 32, 36, 32)])
def test_upsample_cat_conv3x3_x3(case):
    """cfp_upsample_cat_conv3x3 in the default numerics (float32 tensors, f16x3 matrix math; round 5, conv3x3_chunk_x3_kernel<.., UP>): bilinear
    upsample (align_corners=True) of the low-resolution tensor + concatenation with the skip tensor + conv3x3 + folded BatchNorm + LeakyReLU
    (decoder.py:51-58) in ONE launch against float64 torch, and against the pair it replaces (cfp_resize_bilinear + the f16x3 conv on the
    materialised concatenation).  Skip channel counts that are not multiples of 32 (zero-padded chunks), output sizes that are not multiples
    of the 8 x 16 pixel tile, the skip tensor as a column slice of a wider buffer, several channel blocks per tile (Cout = 256)."""
    B, Hs, Ws, H, W, Cup, Csk, Cout = case
    low = rnd(B, Cup, Hs, Ws, seed=1)
    skip = rnd(B, Csk, H, W, seed=2)
    w = rnd(Cout, Cup + Csk, 3, 3, seed=3, scale=1.0 / math.sqrt(9 * (Cup + Csk)))
    scale, shift = rnd(Cout, seed=4).abs() + 0.5, rnd(Cout, seed=5)
    up = F.interpolate(low.double(), size=(H, W), mode="bilinear", align_corners=True)
    ref = F.conv2d(torch.cat([up, skip.double()], 1), w.double(), None, 1, 1)
    ref = F.leaky_relu(ref * scale.double()[None, :, None, None] + shift.double()[None, :, None, None], 0.01)
    la = to_act(nhwc(low), torch.float32, ld=Cup + 8, c0=4)
    cat = ops.new_act(B * H * W, Cup + Csk, torch.float32, DEV, zero=True)       # the concatenation buffer: the skip lives in its right slice
    cat.buf[:, Cup:] = nhwc(skip).to(DEV)
    sk = cat.slice(Cup, Csk)
    wcat = ops.pack_w_x3_cat(w.permute(0, 2, 3, 1).contiguous().to(DEV), Cup)
    out = ops.new_act(B * H * W, Cout, torch.float32, DEV, ld=Cout + 4)
    ops.upsample_cat_conv3x3(la, Hs, Ws, sk, wcat, scale.to(DEV), shift.to(DEV), out, B, H, W, hip.ACT_LRELU, x3=True)
    torch.cuda.synchronize()
    got = from_nhwc(out.torch(), B, H, W)
    _x3_close(got, ref, f"upsample+cat+conv3x3 x3 {case}")
    # the pair it replaces
    ops.resize_bilinear(la, Hs, Ws, (0, 0, Hs, Ws), cat.slice(0, Cup), H, W, (0, 0, H, W), B)
    w32 = w.permute(0, 2, 3, 1).reshape(Cout, 9 * (Cup + Csk)).contiguous().to(DEV)
    out2 = ops.new_act(B * H * W, Cout, torch.float32, DEV)
    ops.conv2d(cat, ops.pack_w_x3(w32), scale.to(DEV), shift.to(DEV), out2, B, H, W, 3, 3, 1, 1, 1, H, W, hip.ACT_LRELU)
    torch.cuda.synchronize()
    _x3_close(from_nhwc(out2.torch(), B, H, W), ref, f"resize + conv x3 {case}")
    _x3_close(got, from_nhwc(out2.torch(), B, H, W).double(), f"fused vs pair {case}", tol=2e-6)


@pytest.mark.parametrize("Co", [128, 64, 32])
def test_conv2d_x3_layernorm_inside_the_split_k_reduce(Co):
    """The global attention's patch convolution (twins.py: sr conv with kernel = stride = the window, then nn.LayerNorm) in the default numerics:
    a few rows x K of thousands, so K is split -- and the LayerNorm (+ residual after it) runs inside the finishing sum of the splits instead of a
    launch of its own.  Against float64, for the automatic plan and forced 2 / 4 / 8 splits, ragged row counts, a padded output pitch."""
    lib = hip.load()
    for (B, H, W, k) in ((1, 36, 48, 12), (2, 27, 45, 9), (1, 30, 40, 6)):
        Cin = Co
        Ho, Wo = H // k, W // k
        x = rnd(B, Cin, H, W, seed=21)
        w = rnd(Co, Cin, k, k, seed=22, scale=1.0 / math.sqrt(Cin * k * k))
        sc, sh = rnd(Co, seed=23).abs() + 0.5, rnd(Co, seed=24)
        g, bt = rnd(Co, seed=25).abs() + 0.5, rnd(Co, seed=26)
        res = rnd(B * Ho * Wo, Co, seed=27)
        y = F.conv2d(x.double(), w.double(), None, k).permute(0, 2, 3, 1).reshape(-1, Co) * sc.double() + sh.double()
        ref = F.layer_norm(y, (Co,), g.double(), bt.double(), 1e-5) + res.double()
        xa = to_act(nhwc(x), torch.float32)
        wx = ops.pack_w_x3(w.permute(0, 2, 3, 1).reshape(Co, k * k * Cin).contiguous().to(DEV))
        M = B * Ho * Wo
        try:
            for splits in (-1, 2, 4, 8):
                lib.cfp_debug_set(1, splits)
                _, sp = ops.conv2d_plan(M, Co, k * k * Cin, hip.F32X3, 0, B, k, k)
                # a plain workspace, and one with the ticket area in front (CFP_CONV_WS_TICKETS): the LayerNorm keeps the reduce launch, the area stays zero
                for ws in (torch.empty(8 * M * Co, device=DEV), ops.ticket_ws(8 * M * Co * 4, DEV)):
                    out = ops.new_act(M, Co, torch.float32, DEV, ld=Co + 8, zero=True)
                    ops.conv2d(xa, wx, sc.to(DEV), sh.to(DEV), out, B, H, W, k, k, k, 0, 0, Ho, Wo, hip.ACT_NONE, to_act(res, torch.float32), ws,
                               ln=(g.to(DEV), bt.to(DEV), 1e-5))
                    torch.cuda.synchronize()
                    assert splits > 0 or sp > 1, "the automatic plan is expected to split this problem"
                    _x3_close(out.torch().cpu(), ref, f"x3 sr conv + LayerNorm in the reduce (Cout {Co}, {B}x{H}x{W} k{k}, splits {splits})", tol=2e-5)
                    assert float(out.buf[:, Co:].abs().max()) == 0
                    if getattr(ws, "cfp_tickets", False):
                        assert int(ws[: hip.CONV_TICKET_BYTES // 4].view(torch.int32).abs().max()) == 0
        finally:
            lib.cfp_debug_set(1, -1)


def test_conv2d_x3_per_image_weights_and_layernorm():
    """Per-image pre-split weights (the squeeze-excite fold) and the LayerNorm that follows as a second kernel."""
    B, HW, Cin, Cout = 3, 300, 232, 128
    x = rnd(B * HW, Cin, seed=1)
    w = rnd(B, Cout, Cin, seed=2, scale=1 / math.sqrt(Cin))
    xa = to_act(x, torch.float32)
    wx = ops.pack_w_x3(w.reshape(B * Cout, Cin).to(DEV)).reshape(B, Cout, -1)
    out = ops.new_act(B * HW, Cout, torch.float32, DEV)
    ops.conv2d(xa, wx, None, None, out, B, 1, HW, 1, 1, 1, 0, 0, 1, HW, hip.ACT_NONE, None, None, per_image_weights=True)
    ref = torch.einsum("bmk,bnk->bmn", x.reshape(B, HW, Cin).double(), w.double()).reshape(B * HW, Cout)
    _x3_close(out.torch().cpu(), ref, "x3 per-image weights")
    # LayerNorm fused into the epilogue (Cout 128 / 64 / 32 / 16: a four-row-wave tile spans the row) and as a second kernel (knob 26 = 0)
    lib = hip.load()
    for Co, rows in ((128, 300), (128, 40000), (64, 300), (64, 130000), (32, 5000), (16, 777)):
        xr = rnd(rows, Cin, seed=11)
        w1 = rnd(Co, Cin, seed=12, scale=1 / math.sqrt(Cin))
        g, bt = rnd(Co, seed=5).abs() + 0.5, rnd(Co, seed=6)
        sc, sh = rnd(Co, seed=8).abs() + 0.5, rnd(Co, seed=9)
        res = rnd(rows, Co, seed=7)
        y = F.relu((xr.double() @ w1.double().t()) * sc.double() + sh.double())
        ref2 = F.layer_norm(y, (Co,), g.double(), bt.double(), 1e-5) + res.double()
        got = []
        for fused in (1, 0):
            lib.cfp_debug_set(26, fused)
            try:
                out2 = ops.new_act(rows, Co, torch.float32, DEV, ld=Co + 8)
                ops.linear(to_act(xr, torch.float32), ops.pack_w_x3(w1.contiguous().to(DEV)), sc.to(DEV), sh.to(DEV), out2, rows, hip.ACT_RELU,
                           to_act(res, torch.float32), None, (g.to(DEV), bt.to(DEV), 1e-5))
                torch.cuda.synchronize()
            finally:
                lib.cfp_debug_set(26, 1)
            _x3_close(out2.torch().cpu(), ref2, f"x3 + LayerNorm + residual (Cout {Co}, rows {rows}, fused {fused})", tol=2e-5)
            got.append(out2.torch().clone())
        assert float((got[0] - got[1]).abs().max()) <= 1e-5 * float(ref2.abs().max())


def test_se_gate_fold_writes_x3_operands():
    """cfp_se_gate_fold with dtype CFP_F32X3: the per-image folded project weights arrive pre-split, equal to packing the float32 fold."""
    B, Cout, C, R, ns = 2, 40, 136, 8, 3
    part = rnd(B * ns * C, seed=1).abs().to(DEV)
    wr, br = rnd(R, C, seed=2, scale=0.1).to(DEV), rnd(R, seed=3).to(DEV)
    we_t, be = rnd(R, C, seed=4).to(DEV), rnd(C, seed=5).to(DEV)
    w = rnd(Cout, C, seed=6, scale=0.1).to(DEV)
    w32 = torch.empty(B, Cout, C, device=DEV)
    ops.se_gate_fold(part, ns, 0.01, wr, br, we_t, be, w, w32, B, Cout, C, R)
    wx = torch.zeros(B, Cout, (C + 31) // 32 * 64, dtype=torch.float16, device=DEV)
    ops.se_gate_fold(part, ns, 0.01, wr, br, we_t, be, w, wx, B, Cout, C, R)
    want = ops.pack_w_x3(w32.reshape(B * Cout, C)).reshape(B, Cout, -1)
    torch.cuda.synchronize()
    assert torch.equal(wx, want)


X3_HALO_CASES = [
    (1, 12, 16, 40, 16, 3, 1, (1, 1, 1, 1)),      # 40 -> 16, exact 16-column tiles, K = 360 ends inside a 32-deep step
    (2, 21, 35, 16, 16, 3, 1, (1, 1, 1, 1)),      # ragged rows and columns
    (1, 19, 30, 8, 32, 3, 1, (1, 1, 1, 1)),       # one channel group per pixel: a K-step spans four taps
    (2, 17, 18, 40, 160, 3, 1, (1, 1, 1, 1)),     # stage 3 expand
    (1, 9, 33, 56, 224, 3, 1, (1, 1, 1, 1)),      # stage 4 expand
    (1, 16, 16, 64, 64, 3, 1, (1, 1, 1, 1)),
    (1, 25, 20, 32, 128, 3, 1, (0, 2, 2, 0)),     # asymmetric padding (conv0's channel counts)
    (3, 8, 8, 24, 40, 3, 1, (1, 1, 1, 1)),        # channel counts that fill no tile exactly
    (1, 5, 70, 48, 96, 3, 1, (1, 1, 1, 1)),       # fewer rows than a tile
    (1, 18, 34, 128, 128, 3, 1, (1, 1, 1, 1)),    # the depth head's conv (8 x 16 pixel tiles only: the halo is 95 KB)
    (1, 10, 20, 168, 64, 3, 1, (1, 1, 1, 1)),     # up3's first conv: the deepest halo that fits
    (2, 11, 13, 80, 32, 3, 1, (1, 1, 1, 1)),      # up4's first conv
]
X3_HALO_CAP = [16, 32, 64, 64, 128, 160, 224, 32, 128, 80]      # output channels per workgroup of each tile (csrc/conv3x3_halo_x3.hip kHCfg)


X3_CHUNK_CASES = [
    (1, 18, 34, 128, 128, 3, 1, (1, 1, 1, 1)),    # the depth head's conv: four chunks
    (2, 21, 35, 64, 32, 3, 1, (1, 1, 1, 1)),      # two chunks, ragged rows and columns
    (1, 9, 33, 64, 64, 3, 1, (1, 1, 1, 1)),
    (1, 25, 20, 32, 128, 3, 1, (0, 2, 2, 0)),     # one chunk (no pipelining), asymmetric padding
    (1, 16, 16, 256, 256, 3, 1, (1, 1, 1, 1)),    # eight chunks, two channel blocks for the 128-wide tiles
    (3, 8, 8, 96, 40, 3, 1, (1, 1, 1, 1)),        # three chunks, channel count that fills no tile exactly
    (1, 5, 70, 160, 96, 3, 1, (1, 1, 1, 1)),      # five chunks, fewer rows than a tile
]


@pytest.mark.parametrize("variant", [20, 21, 22, 23, 24, 25, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 43, 44, 45, 46])
def test_conv3x3_chunk_x3_every_variant(variant):
    """The chunk-pipelined f16x3 3x3 kernel for deep inputs (Cin % 32 == 0: the halo by 32-channel chunks, double buffered, K loop chunk-major):
    every tile forced through the debug knob (500 + v) against a float64 reference and the f16x3 implicit GEMM.  Variants 30-35 (round 5) =
    20-25 with three weight stages and counted waits: the same products in the same order -> bit-identical to 20-25, ten launches in a row;
    36-38 = one chunk buffer (two workgroups per CU for the 128-channel tile): bit-identical to 23 / 25 / 21."""
    lib = hip.load()
    try:
        for case in X3_CHUNK_CASES:
            B, H, W, Cin, Cout, k, s, pads = case
            ref, xa, w32, wx, scale, shift, ra, Ho, Wo = _x3_problem(case)
            lib.cfp_debug_set(0, 500 + variant)
            out = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV, ld=Cout + 24, zero=True)
            out = ops.Act(out.buf, 16, Cout)
            ops.conv2d(xa, wx, scale, shift, out, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
            torch.cuda.synchronize()
            _x3_close(from_nhwc(out.torch(), B, Ho, Wo), ref, f"x3 chunk v{variant} conv {case}")
            assert float(out.buf[:, :16].abs().max()) == 0 and float(out.buf[:, 16 + Cout:].abs().max()) == 0
            lib.cfp_debug_set(0, 413)
            out2 = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV)
            ops.conv2d(xa, wx, scale, shift, out2, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
            torch.cuda.synchronize()
            assert float((out2.torch() - out.torch()).abs().max()) <= 2e-6 * float(ref.abs().max())
            if variant >= 30:
                lib.cfp_debug_set(0, 500 + ({36: 23, 37: 25, 38: 21, 39: 23, 43: 21, 44: 24, 45: 24, 46: 20}.get(variant, variant - 10)))
                ops.conv2d(xa, wx, scale, shift, out2, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
                lib.cfp_debug_set(0, 500 + variant)
                for _ in range(10):
                    ops.conv2d(xa, wx, scale, shift, out, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
                    torch.cuda.synchronize()
                    assert torch.equal(out.torch(), out2.torch()), (variant, case)
    finally:
        lib.cfp_debug_set(0, -1)


@pytest.mark.parametrize("variant", list(range(10)) + [99])
def test_conv3x3_halo_x3_every_variant(variant):
    """The f16x3 whole-depth-halo 3x3 kernel (float32 tensors, halo split once into LDS, the implicit GEMM's pre-split weights), every tile
    forced through the debug knob (500 + v; 599 = its automatic tile) against a float64 reference, and equal to the f16x3 implicit GEMM to
    float32 re-association."""
    lib = hip.load()
    ran = 0
    try:
        for case in X3_HALO_CASES:
            B, H, W, Cin, Cout, k, s, pads = case
            ref, xa, w32, wx, scale, shift, ra, Ho, Wo = _x3_problem(case)
            lib.cfp_debug_set(0, 500 + variant)
            out = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV, ld=Cout + 24, zero=True)
            out = ops.Act(out.buf, 16, Cout)
            try:
                ops.conv2d(xa, wx, scale, shift, out, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
            except RuntimeError as e:
                assert "cannot run this problem" in str(e) and variant != 99      # halo too deep for this tile's LDS
                continue
            torch.cuda.synchronize()
            ran += 1
            _x3_close(from_nhwc(out.torch(), B, Ho, Wo), ref, f"x3 halo v{variant} conv {case}")
            assert float(out.buf[:, :16].abs().max()) == 0 and float(out.buf[:, 16 + Cout:].abs().max()) == 0
            lib.cfp_debug_set(0, 413)
            out2 = ops.new_act(B * Ho * Wo, Cout, torch.float32, DEV)
            ops.conv2d(xa, wx, scale, shift, out2, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
            torch.cuda.synchronize()
            assert float((out2.torch() - out.torch()).abs().max()) <= 2e-6 * float(ref.abs().max())
    finally:
        lib.cfp_debug_set(0, -1)
    assert ran >= (len(X3_HALO_CASES) if variant == 99 else 5)


def test_conv3x3_halo_x3_is_what_the_plan_picks_for_many_pixel_layers():
    v, sp = ops.conv2d_plan(8 * 240 * 320, 128, 9 * 32, hip.F32X3, 0, 8, 3, 1)
    assert v == 500 and sp == 1
    v, _ = ops.conv2d_plan(8 * 15 * 20, 256, 9 * 392, hip.F32X3, 0, 8, 3, 1)
    assert 400 <= v < 500


DIRECT3_CASES = [
    (1, 12, 16, 40, 16, 3, 1, (1, 1, 1, 1)),      # Cin < 64 (one partly filled channel chunk), Cout 16
    (1, 15, 20, 392, 256, 3, 1, (1, 1, 1, 1)),    # 7 channel chunks, tail chunk of 8 channels, 2 N-tiles
    (1, 13, 17, 32, 32, 3, 1, (1, 1, 1, 1)),      # ragged patch edges
    (2, 40, 50, 80, 32, 3, 1, (1, 1, 1, 1)),      # up4-like: chunk 64 + 16
    (1, 33, 47, 168, 64, 3, 1, (0, 2, 2, 0)),     # asymmetric padding
    (2, 16, 32, 128, 128, 3, 1, (1, 1, 1, 1)),    # exact tiles
]


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("variant", list(range(6)))
def test_conv3x3_direct_every_variant(variant, dtype):
    """The LDS-halo direct 3x3 kernel, every tile variant forced through the debug knob."""
    lib = hip.load()
    try:
        lib.cfp_debug_set(0, 200 + variant)
        for case in DIRECT3_CASES:
            B, H, W, Cin, Cout, k, s, pads = case
            ref, xa, wa, scale, shift, ra, Ho, Wo = _conv_ref_and_args(case, dtype)
            v, sp = ops.conv2d_plan(B * Ho * Wo, Cout, 9 * Cin, hip.BF16, 0, B, 3, 1)
            assert v == 200 + variant
            out = ops.new_act(B * Ho * Wo, Cout, dtype, DEV, ld=Cout + 24, zero=True)
            out = ops.Act(out.buf, 16, Cout)
            ops.conv2d(xa, wa, scale, shift, out, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
            torch.cuda.synchronize()
            close(from_nhwc(out.torch(), B, Ho, Wo), ref, dtype, f"direct3x3 v{variant} conv {case}")
            assert float(out.buf[:, :16].abs().max()) == 0 and float(out.buf[:, 16 + Cout:].abs().max()) == 0
    finally:
        lib.cfp_debug_set(0, -1)


HALO_CAP = [16, 32, 64, 64, 128, 160, 224, 32]          # output channels per workgroup of each conv3x3_halo variant (csrc/conv3x3_halo.hip kHCfg); wider layers take several channel blocks
HALO_CASES = [
    (1, 12, 16, 40, 16, 3, 1, (1, 1, 1, 1)),      # encoder stage 1 (40 -> 16), exact 16-column tiles
    (2, 21, 35, 16, 16, 3, 1, (1, 1, 1, 1)),      # 16 -> 16 with skip, ragged rows and columns, even chunk count (padded pixel pitch)
    (1, 19, 30, 8, 32, 3, 1, (1, 1, 1, 1)),       # one chunk per pixel: four taps per 32-deep MFMA step
    (2, 17, 18, 40, 160, 3, 1, (1, 1, 1, 1)),     # stage 3 expand (40 -> 160): K = 360 ends inside a 64-deep step
    (1, 9, 33, 56, 224, 3, 1, (1, 1, 1, 1)),      # stage 4 expand (56 -> 224)
    (1, 16, 16, 64, 64, 3, 1, (1, 1, 1, 1)),      # decoder, K = 576 = nine whole steps
    (1, 25, 20, 32, 128, 3, 1, (0, 2, 2, 0)),     # asymmetric padding
    (3, 8, 8, 24, 40, 3, 1, (1, 1, 1, 1)),        # channel counts that fill no tile exactly
    (1, 5, 70, 48, 96, 3, 1, (1, 1, 1, 1)),       # fewer rows than a tile
]


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("variant", list(range(8)))
def test_conv3x3_halo_every_variant(variant, dtype):
    """The whole-depth halo 3x3 kernel (few input channels, many pixels), every tile variant forced through the debug knob, against
    float32 torch; the automatic plan must pick it for big problems of this family and the result must not depend on the tile."""
    lib = hip.load()
    ran = 0
    try:
        for case in HALO_CASES:
            B, H, W, Cin, Cout, k, s, pads = case
            if Cout > 4 * HALO_CAP[variant]:
                continue
            ref, xa, wa, scale, shift, ra, Ho, Wo = _conv_ref_and_args(case, dtype)
            lib.cfp_debug_set(0, 300 + variant)
            out = ops.new_act(B * Ho * Wo, Cout, dtype, DEV, ld=Cout + 24, zero=True)
            out = ops.Act(out.buf, 16, Cout)
            ops.conv2d(xa, wa, scale, shift, out, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
            torch.cuda.synchronize()
            close(from_nhwc(out.torch(), B, Ho, Wo), ref, dtype, f"halo3x3 v{variant} conv {case}")
            assert float(out.buf[:, :16].abs().max()) == 0 and float(out.buf[:, 16 + Cout:].abs().max()) == 0
            # against the implicit GEMM on the same 16-bit operands: float32 re-association only
            lib.cfp_debug_set(0, 4)
            out2 = ops.new_act(B * Ho * Wo, Cout, dtype, DEV)
            ops.conv2d(xa, wa, scale, shift, out2, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
            torch.cuda.synchronize()
            a, b2 = out.torch().float(), out2.torch().float()
            ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
            assert float(((a - b2).abs() / b2.abs().clamp(min=1.0)).max()) <= 2 * ulp, f"halo v{variant} vs igemm2 {case}"
            assert float((a != b2).float().mean()) < 0.02
            ran += 1
    finally:
        lib.cfp_debug_set(0, -1)
    assert ran > 0
    assert ops.conv2d_plan(8 * 120 * 160, 160, 360, hip.BF16, 0, 8, 3, 1)[0] == 300
    assert ops.conv2d_plan(8 * 240 * 320, 16, 360, hip.BF16, 0, 8, 3, 1)[0] == 300
    assert ops.conv2d_plan(8 * 15 * 20, 160, 360, hip.BF16, 0, 8, 3, 1)[0] != 300        # too few pixels: implicit GEMM
    assert ops.conv2d_plan(8 * 120 * 160, 160, 9 * 72, hip.BF16, 0, 8, 3, 1)[0] != 300      # too many input channels


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 7])
def test_conv3x3_halo_stride2(variant, dtype):
    """The halo kernel's stride-2 form (stem, first block of an encoder stage; TF-"same" padding = 0 top / left, 1 bottom / right on even
    inputs): against float32 torch, against the implicit GEMM on the same operands, and bit-exact on small integers."""
    lib = hip.load()
    cases = [(1, 34, 46, 8, 40, 3, 2, (0, 0, 1, 1)), (2, 24, 32, 16, 64, 3, 2, (0, 0, 1, 1)), (1, 21, 35, 40, 160, 3, 2, (1, 1, 1, 1)),
             (1, 66, 20, 24, 16, 3, 2, (0, 0, 1, 1)), (1, 10, 130, 64, 32, 3, 2, (1, 0, 0, 1))]
    ran = 0
    try:
        for case in cases:
            B, H, W, Cin, Cout, k, s, pads = case
            if Cout > 4 * HALO_CAP[variant]:
                continue
            th = 16 if variant in (0, 1, 2) else 8
            slots = Cin // 8 + (1 - (Cin // 8) % 2)
            if ((th - 1) * 2 + 3) * 33 * slots * 16 + 2 * min(HALO_CAP[variant], 224) * 128 > 160 * 1024:
                continue                                    # the stride-2 halo of this tile does not fit the LDS: the plan never picks it
            ref, xa, wa, scale, shift, ra, Ho, Wo = _conv_ref_and_args(case, dtype)
            lib.cfp_debug_set(0, 300 + variant)
            out = ops.new_act(B * Ho * Wo, Cout, dtype, DEV, ld=Cout + 24, zero=True)
            out = ops.Act(out.buf, 16, Cout)
            ops.conv2d(xa, wa, scale, shift, out, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
            torch.cuda.synchronize()
            close(from_nhwc(out.torch(), B, Ho, Wo), ref, dtype, f"halo3x3 stride 2 v{variant} conv {case}")
            assert float(out.buf[:, :16].abs().max()) == 0 and float(out.buf[:, 16 + Cout:].abs().max()) == 0
            lib.cfp_debug_set(0, 4)
            out2 = ops.new_act(B * Ho * Wo, Cout, dtype, DEV)
            ops.conv2d(xa, wa, scale, shift, out2, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_SILU, ra, None)
            torch.cuda.synchronize()
            assert torch.equal(out.torch().view(torch.int16), out2.torch().view(torch.int16)), f"halo stride 2 v{variant} != igemm2 {case}"      # same K order
            # integers
            x = _int_tensor((B, Cin, H, W), -3, 3, 1)
            w = _int_tensor((Cout, Cin, k, k), -2, 2, 2)
            refi = F.conv2d(F.pad(x.double(), (pads[1], pads[3], pads[0], pads[2])), w.double(), None, s).float()
            lib.cfp_debug_set(0, 300 + variant)
            outi = ops.new_act(B * Ho * Wo, Cout, dtype, DEV)
            wi = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dtype).to(DEV)
            ops.conv2d(to_act(nhwc(x), dtype), wi, None, None, outi, B, H, W, k, k, s, pads[0], pads[1], Ho, Wo, hip.ACT_NONE, None, None)
            torch.cuda.synchronize()
            got = outi.torch().cpu().reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
            assert torch.equal(got.view(torch.int16), _bits(refi, dtype)), f"halo stride 2 variant {variant} integers {case}"
            ran += 1
    finally:
        lib.cfp_debug_set(0, -1)
    assert ran > 0


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("variant", list(range(8)))
def test_conv3x3_halo_bit_exact_on_integers(variant, dtype):
    lib = hip.load()
    cases = [(2, 20, 24, 40, 48, 3, 1, (1, 1, 1, 1)), (1, 17, 33, 16, 16, 3, 1, (1, 1, 1, 1)), (1, 10, 50, 64, 32, 3, 1, (1, 1, 1, 1)),
             (1, 13, 19, 8, 160, 3, 1, (1, 1, 1, 1)), (1, 11, 16, 56, 224, 3, 1, (0, 1, 1, 0))]
    ran = 0
    try:
        lib.cfp_debug_set(0, 300 + variant)
        for case in cases:
            B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
            if Cout > 4 * HALO_CAP[variant]:
                continue
            x = _int_tensor((B, Cin, H, W), -3, 3, 1)
            w = _int_tensor((Cout, Cin, k, k), -2, 2, 2)
            Ho, Wo = (H + pt + pb - k) // s + 1, (W + pl + pr - k) // s + 1
            ref = F.conv2d(F.pad(x.double(), (pl, pr, pt, pb)), w.double(), None, s).float()
            out = ops.new_act(B * Ho * Wo, Cout, dtype, DEV)
            wa = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dtype).to(DEV)
            ops.conv2d(to_act(nhwc(x), dtype), wa, None, None, out, B, H, W, k, k, s, pt, pl, Ho, Wo, hip.ACT_NONE, None, None)
            torch.cuda.synchronize()
            got = out.torch().cpu().reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
            assert torch.equal(got.view(torch.int16), _bits(ref, dtype)), f"halo variant {variant} case {case}"
            ran += 1
    finally:
        lib.cfp_debug_set(0, -1)
    assert ran > 0


@pytest.mark.parametrize("dtype", HALF)
def test_conv2d_gen2_matches_gen1_bitwise_inputs(dtype):
    """Same bf16 inputs through both kernel generations: results agree to bf16 rounding of an f32
    accumulation in a different order."""
    lib = hip.load()
    case = (2, 30, 40, 168, 64, 3, 1, (1, 1, 1, 1))
    B, H, W, Cin, Cout, k, s, _ = case
    ref, xa, wa, scale, shift, ra, Ho, Wo = _conv_ref_and_args(case, dtype)
    outs = []
    try:
        for v1 in (0, 1):
            lib.cfp_debug_set(2, v1)
            out = ops.new_act(B * Ho * Wo, Cout, dtype, DEV)
            ops.conv2d(xa, wa, scale, shift, out, B, H, W, k, k, s, 1, 1, Ho, Wo, hip.ACT_SILU, ra, None)
            torch.cuda.synchronize()
            outs.append(out.torch().float().cpu())
    finally:
        lib.cfp_debug_set(2, 0)
    assert float((outs[0] - outs[1]).abs().max()) <= 2.0 ** -6 * float(outs[0].abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,Cin,Cout", [(300, 256, 128), (4800, 128, 64), (1000, 64, 32), (77, 64, 16), (50, 32, 256)])
def test_linear_with_fused_layernorm(rows, Cin, Cout, dtype):
    """out = LN(x @ W^T) * g + b + residual  (transformer.py:63,68-70).  Cout = 256 has no exact-width
    tile and exercises the two-kernel fallback."""
    x = q(rnd(rows, Cin, seed=1), dtype)
    w = q(rnd(Cout, Cin, seed=2, scale=1.0 / math.sqrt(Cin)), dtype)
    g, b = rnd(Cout, seed=3).abs() + 0.5, rnd(Cout, seed=4)
    res = q(rnd(rows, Cout, seed=5), dtype)
    y = x @ w.t()
    if dtype in HALF:
        y = q(y, dtype)           # the pre-LayerNorm tile is rounded to the storage type
    ref = F.layer_norm(y, (Cout,), g, b, 1e-5) + res
    out = ops.new_act(rows, Cout, dtype, DEV, ld=Cout + 8, zero=True)
    out = ops.Act(out.buf, 8, Cout)
    ops.linear(to_act(x, dtype), w.to(dtype).to(DEV), None, None, out, rows, hip.ACT_NONE, to_act(res, dtype), None,
               ln=(g.to(DEV), b.to(DEV), 1e-5))
    torch.cuda.synchronize()
    close(out.torch().float().cpu(), ref, dtype, f"linear+LN {rows}x{Cin}->{Cout}")
    assert float(out.buf[:, :8].abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,Cin,Cout", [(3, 300, 1392, 232), (2, 1200, 816, 136), (2, 70, 64, 24)])
def test_pointwise_conv_per_image_weights(B, HW, Cin, Cout, dtype):
    """Image b of the batch multiplies with its own weight matrix (SE gate folded into the project conv)."""
    x = q(rnd(B, HW, Cin, seed=1), dtype)
    w = q(rnd(B, Cout, Cin, seed=2, scale=1.0 / math.sqrt(Cin)), dtype)
    scale, shift = rnd(Cout, seed=3).abs() + 0.5, rnd(Cout, seed=4)
    res = q(rnd(B, HW, Cout, seed=5), dtype)
    ref = torch.einsum("bmk,bnk->bmn", x, w) * scale + shift + res
    out = ops.new_act(B * HW, Cout, dtype, DEV)
    ops.conv2d(to_act(x.reshape(B * HW, Cin), dtype), w.to(dtype).to(DEV).contiguous(), scale.to(DEV), shift.to(DEV), out,
               B, 1, HW, 1, 1, 1, 0, 0, 1, HW, hip.ACT_NONE, to_act(res.reshape(B * HW, Cout), dtype), None,
               per_image_weights=True)
    torch.cuda.synchronize()
    close(out.torch().float().cpu().reshape(B, HW, Cout), ref, dtype, f"per-image weights {B}x{HW}x{Cin}->{Cout}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", [hip.ACT_NONE, hip.ACT_RELU, hip.ACT_LRELU, hip.ACT_GELU, hip.ACT_SIGMOID])
def test_conv_activations_no_scale(act, dtype):
    x = q(rnd(1, 64, 6, 10, seed=7), dtype)
    w = q(rnd(96, 64, 1, 1, seed=8, scale=0.2), dtype)
    ref = F.conv2d(x, w)
    ref = {hip.ACT_NONE: lambda t: t, hip.ACT_RELU: F.relu, hip.ACT_LRELU: lambda t: F.leaky_relu(t, 0.01),
           hip.ACT_GELU: F.gelu, hip.ACT_SIGMOID: torch.sigmoid}[act](ref)
    out = ops.new_act(60, 96, dtype, DEV)
    ops.conv2d(to_act(nhwc(x), dtype), w.reshape(96, 64).to(dtype).to(DEV), None, None, out, 1, 6, 10, 1, 1, 1, 0, 0, 6, 10, act)
    close(from_nhwc(out.torch(), 1, 6, 10), ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 30, 40, 224, 2, (0, 0, 1, 1)), (1, 15, 20, 1392, 1, (1, 1, 1, 1)),
                                  (2, 9, 7, 64, 2, (1, 1, 1, 1)), (1, 5, 3, 8, 1, (1, 1, 1, 1))])
def test_dwconv3x3(case, dtype):
    B, H, W, Cc, s, (pt, pl, pb, pr) = case
    x = q(rnd(B, Cc, H, W, seed=1), dtype)
    w = q(rnd(Cc, 1, 3, 3, seed=2, scale=0.4), dtype)
    scale, shift = rnd(Cc, seed=3).abs() + 0.5, rnd(Cc, seed=4)
    Ho, Wo = (H + pt + pb - 3) // s + 1, (W + pl + pr - 3) // s + 1
    ref = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, None, s, 0, 1, Cc)
    ref = F.silu(ref * scale[None, :, None, None] + shift[None, :, None, None])
    wa = w.reshape(Cc, 9).t().contiguous().to(dtype).to(DEV)       # [9][C]
    out = ops.new_act(B * Ho * Wo, Cc, dtype, DEV)
    ops.dwconv3x3(to_act(nhwc(x), dtype), wa, scale.to(DEV), shift.to(DEV), out, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
    close(from_nhwc(out.torch(), B, Ho, Wo), ref, dtype, f"dw3x3 {case}")


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", [(1, 30, 40, 128, 7), (2, 60, 80, 64, 15), (1, 120, 160, 32, 31), (2, 33, 47, 16, 31), (1, 9, 9, 8, 15)])
def test_dwconv_large_mfma_toeplitz(case, dtype):
    """Large-kernel depthwise as banded Toeplitz GEMMs on the matrix cores (bf16) vs F.conv2d."""
    B, H, W, Cc, k = case
    x = q(rnd(B, Cc, H, W, seed=1), dtype)
    w = q(rnd(Cc, 1, k, k, seed=2, scale=1.0 / k), dtype)         # the band table stores bf16 weights
    scale, shift = rnd(Cc, seed=3).abs() + 0.5, rnd(Cc, seed=4)
    ref = F.relu(F.conv2d(x, w, None, 1, (k - 1) // 2, 1, Cc) * scale[None, :, None, None] + shift[None, :, None, None])
    tb = ops.toeplitz_bands(w, dtype).to(DEV)
    out = ops.new_act(B * H * W, Cc, dtype, DEV, ld=Cc + 8, zero=True)
    out = ops.Act(out.buf, 8, Cc)
    ops.dwconv_large_mfma(to_act(nhwc(x), dtype, ld=Cc + 16, c0=8), tb, scale.to(DEV), shift.to(DEV), out, B, H, W, k, hip.ACT_RELU)
    torch.cuda.synchronize()
    close(from_nhwc(out.torch(), B, H, W), ref, dtype, f"dwlarge mfma {case}")
    assert float(out.buf[:, :8].abs().max()) == 0


@pytest.mark.parametrize("case", [(1, 70, 45, 32, 31), (2, 33, 40, 64, 15), (1, 30, 40, 128, 7), (2, 17, 19, 12, 31), (1, 64, 32, 4, 15)])
def test_dwconv_large_x3_toeplitz(case):
    """The large-kernel depthwise convolution in the default numerics: float32 tensors, banded-Toeplitz GEMMs with every operand as hi + lo
    halves (three MFMAs per block) against float64 F.conv2d; ragged patches, channel counts of 4 ... 128."""
    B, H, W, Cc, k = case
    x = rnd(B, Cc, H, W, seed=1)
    w = rnd(Cc, 1, k, k, seed=2, scale=1.0 / k)
    scale, shift = rnd(Cc, seed=3).abs() + 0.5, rnd(Cc, seed=4)
    ref = F.relu(F.conv2d(x.double(), w.double(), None, 1, (k - 1) // 2, 1, Cc) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
    tb = ops.toeplitz_bands_x3(w).to(DEV)
    out = ops.new_act(B * H * W, Cc, torch.float32, DEV, ld=Cc + 8, zero=True)
    out = ops.Act(out.buf, 4, Cc)
    ops.dwconv_large_mfma(to_act(nhwc(x), torch.float32, ld=Cc + 16, c0=8), tb, scale.to(DEV), shift.to(DEV), out, B, H, W, k, hip.ACT_RELU)
    torch.cuda.synchronize()
    _x3_close(from_nhwc(out.torch(), B, H, W), ref, f"dwlarge x3 {case}")
    assert float(out.buf[:, :4].abs().max()) == 0 and float(out.buf[:, 4 + Cc:].abs().max()) == 0
    if k == 31:       # the other tile of k = 31 (cfp_debug_set(30, .): 64 x 32 / 32 x 32 pixels): same products per output -> same bits
        lib = hip.load()
        first = out.torch().clone()
        try:
            lib.cfp_debug_set(30, 0 if int(os.environ.get("CFP_DWL3_DEFAULT", "1")) else 1)
            ops.dwconv_large_mfma(to_act(nhwc(x), torch.float32, ld=Cc + 16, c0=8), tb, scale.to(DEV), shift.to(DEV), out, B, H, W, k, hip.ACT_RELU)
            torch.cuda.synchronize()
            assert torch.equal(out.torch(), first)
        finally:
            lib.cfp_debug_set(30, int(os.environ.get("CFP_DWL3_DEFAULT", "1")))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 30, 40, 224, 2, (0, 0, 1, 1)), (2, 15, 20, 1392, 1, (1, 1, 1, 1)), (3, 30, 40, 816, 1, (1, 1, 1, 1)),
                                  (2, 9, 7, 64, 2, (1, 1, 1, 1)), (1, 5, 3, 8, 1, (1, 1, 1, 1))])
def test_dwconv3x3_with_fused_channel_sums(case, dtype):
    B, H, W, Cc, s, (pt, pl, pb, pr) = case
    x = q(rnd(B, Cc, H, W, seed=1), dtype)
    w = q(rnd(Cc, 1, 3, 3, seed=2, scale=0.4), dtype)
    scale, shift = rnd(Cc, seed=3).abs() + 0.5, rnd(Cc, seed=4)
    Ho, Wo = (H + pt + pb - 3) // s + 1, (W + pl + pr - 3) // s + 1
    ref = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, None, s, 0, 1, Cc)
    ref = F.silu(ref * scale[None, :, None, None] + shift[None, :, None, None])
    wa = w.reshape(Cc, 9).t().contiguous().to(dtype).to(DEV)
    out = ops.new_act(B * Ho * Wo, Cc, dtype, DEV)
    ns = ops.dwconv3x3_strips(B, Ho, Wo, Cc, s, ops.DT[dtype])
    part = torch.full((B, ns, Cc), float("nan"), device=DEV)
    ops.dwconv3x3_sum(to_act(nhwc(x), dtype), wa, scale.to(DEV), shift.to(DEV), out, part, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
    torch.cuda.synchronize()
    got = from_nhwc(out.torch(), B, Ho, Wo)
    close(got, ref, dtype, f"dw3x3+sum {case}")
    sums = part.sum(1).cpu()
    want = ref.sum((2, 3))                      # sums of the activated tensor (taken before the storage rounding)
    assert torch.allclose(sums, want, rtol=1e-3, atol=3e-3 * float(want.abs().max())), float((sums - want).abs().max())
    part2 = torch.zeros_like(part)
    ops.dwconv3x3_sum(to_act(nhwc(x), dtype), wa, scale.to(DEV), shift.to(DEV), out, part2, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
    torch.cuda.synchronize()
    assert torch.equal(part, part2)             # deterministic reduction order


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", [(8, 15, 20, 1392, 1, 58, 232), (8, 30, 40, 816, 1, 34, 136), (2, 30, 40, 448, 1, 28, 112), (8, 60, 80, 224, 2, 14, 112),
                                  (3, 13, 17, 672, 1, 28, 136), (1, 7, 5, 48, 1, 6, 24)])
def test_squeeze_excite_through_the_depthwise_kernel(case, dtype):
    """Round 3's squeeze-excite path: cfp_dwconv3x3_se_nhwc (depthwise 3x3 + BN + SiLU + the reduce FC's partial dot products per
    workgroup) -> cfp_se_gate_fold2 (sum of the partials -> SiLU -> expand FC -> sigmoid -> gate folded into float32 project weights,
    rounded once) against timm's SqueezeExcite written out in torch on the kernel's own depthwise output, and against the round-2 pair
    (cfp_dwconv3x3_sum_nhwc -> cfp_se_gate_fold): same depthwise output bit for bit, same gate to float32 round-off."""
    B, H, W, C, s, R, Cout = case
    Ho, Wo = -(-H // s), -(-W // s)
    pt, pl = max((Ho - 1) * s + 3 - H, 0) // 2, max((Wo - 1) * s + 3 - W, 0) // 2
    x = q(rnd(B, C, H, W, seed=21), dtype)
    wdw = q(rnd(C, 1, 3, 3, seed=22, scale=0.4), dtype)
    scale, shift = (rnd(C, seed=23).abs() + 0.5).to(DEV), rnd(C, seed=24).to(DEV)
    wr, br = rnd(R, C, seed=25, scale=1.0 / math.sqrt(C)).to(DEV), rnd(R, seed=26, scale=0.1).to(DEV)
    we_t, be = rnd(R, C, seed=27, scale=0.3).to(DEV), rnd(C, seed=28, scale=0.1).to(DEV)
    wp = rnd(Cout, C, seed=29, scale=1.0 / math.sqrt(C)).to(DEV)
    wa = wdw.reshape(C, 9).t().contiguous().to(dtype).to(DEV)
    xin = to_act(nhwc(x), dtype)
    out1, out2 = ops.new_act(B * Ho * Wo, C, dtype, DEV), ops.new_act(B * Ho * Wo, C, dtype, DEV)
    K = ops.dwconv3x3_se_parts(B, Ho, Wo, C, s, ops.DT[dtype])
    assert K > 0
    hpart = torch.full((B, K, R), float("nan"), device=DEV)
    ops.dwconv3x3_se(xin, wa, scale, shift, out1, wr, hpart, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
    wb = torch.empty(B, Cout, C, dtype=dtype, device=DEV)
    ops.se_gate_fold2(hpart, K, 1.0 / (Ho * Wo), br, we_t, be, wp, wb, B, Cout, C, R)
    # the round-2 pair on the same input
    ns = ops.dwconv3x3_strips(B, Ho, Wo, C, s, ops.DT[dtype])
    part = torch.empty(B, ns, C, device=DEV)
    ops.dwconv3x3_sum(xin, wa, scale, shift, out2, part, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
    wb_old = torch.empty_like(wb)
    ops.se_gate_fold(part, ns, 1.0 / (Ho * Wo), wr, br, we_t, be, wp.to(dtype), wb_old, B, Cout, C, R)
    torch.cuda.synchronize()
    assert torch.equal(out1.buf.view(torch.int16), out2.buf.view(torch.int16))
    assert torch.isfinite(hpart).all()
    # reference gate from the kernel's own (float32, unrounded) channel sums = part
    mean = part.sum(1) / (Ho * Wo)
    hid = F.silu(mean @ wr.t() + br)
    gate = torch.sigmoid(hid @ we_t + be)
    want = (wp[None] * gate[:, None, :])
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    err = (wb.float() - want).abs()
    assert bool((err <= 1.01 * ulp * want.abs() + 1e-6 * float(want.abs().max())).all()), float((err / want.abs().clamp(min=1e-6)).max())
    # and the old pair agrees up to its double rounding (project weights rounded before the fold)
    assert float((wb.float() - wb_old.float()).abs().max()) <= 3.0 * ulp * float(want.abs().max())


@pytest.mark.parametrize("case", [(8, 15, 20, 1392, 1, 58, 232), (2, 30, 40, 448, 1, 28, 112), (8, 60, 80, 224, 2, 14, 112), (3, 13, 17, 672, 1, 28, 136),
                                  (1, 7, 5, 48, 1, 6, 24)])
def test_squeeze_excite_through_the_float32_depthwise_kernel(case):
    """The same path in float32 storage (the default f16x3 mode): the VALU depthwise kernel leaves the reduce FC's partial dot products,
    cfp_se_gate_fold2 finishes the gate and writes the per-image project weights as pre-split f16x3 operands.  Against the round-2 pair
    (channel sums -> cfp_se_gate_fold): depthwise output bit for bit, the folded operands (hi + lo) to float32 round-off."""
    B, H, W, C, s, R, Cout = case
    dtype = torch.float32
    Ho, Wo = -(-H // s), -(-W // s)
    pt, pl = max((Ho - 1) * s + 3 - H, 0) // 2, max((Wo - 1) * s + 3 - W, 0) // 2
    x = rnd(B, C, H, W, seed=21)
    wdw = rnd(C, 1, 3, 3, seed=22, scale=0.4)
    scale, shift = (rnd(C, seed=23).abs() + 0.5).to(DEV), rnd(C, seed=24).to(DEV)
    wr, br = rnd(R, C, seed=25, scale=1.0 / math.sqrt(C)).to(DEV), rnd(R, seed=26, scale=0.1).to(DEV)
    we_t, be = rnd(R, C, seed=27, scale=0.3).to(DEV), rnd(C, seed=28, scale=0.1).to(DEV)
    wp = rnd(Cout, C, seed=29, scale=1.0 / math.sqrt(C)).to(DEV)
    wa = wdw.reshape(C, 9).t().contiguous().to(DEV)
    xin = to_act(nhwc(x), dtype)
    out1, out2 = ops.new_act(B * Ho * Wo, C, dtype, DEV), ops.new_act(B * Ho * Wo, C, dtype, DEV)
    K = ops.dwconv3x3_se_parts(B, Ho, Wo, C, s, ops.DT[dtype])
    assert K > 0
    hpart = torch.full((B, K, R), float("nan"), device=DEV)
    ops.dwconv3x3_se(xin, wa, scale, shift, out1, wr, hpart, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
    wrow = (C + 31) // 32 * 64
    wb = torch.zeros(B, Cout, wrow, dtype=torch.float16, device=DEV)
    ops.se_gate_fold2(hpart, K, 1.0 / (Ho * Wo), br, we_t, be, wp, wb, B, Cout, C, R, x3=True)
    ns = ops.dwconv3x3_strips(B, Ho, Wo, C, s, ops.DT[dtype])
    part = torch.empty(B, ns, C, device=DEV)
    ops.dwconv3x3_sum(xin, wa, scale, shift, out2, part, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
    wb_old = torch.zeros_like(wb)
    ops.se_gate_fold(part, ns, 1.0 / (Ho * Wo), wr, br, we_t, be, wp, wb_old, B, Cout, C, R)
    torch.cuda.synchronize()
    assert torch.equal(out1.buf, out2.buf)
    assert torch.isfinite(hpart).all()
    mean = part.sum(1) / (Ho * Wo)
    # the partial dot products add up to the reduce FC of the channel sums
    got_h = hpart.sum(1)
    want_h = part.sum(1) @ wr.t()
    assert float((got_h - want_h).abs().max()) <= 2e-5 * float(want_h.abs().max()) + 1e-5
    hid = F.silu(mean @ wr.t() + br)
    gate = torch.sigmoid(hid @ we_t + be)
    want = ops.pack_w_x3((wp[None] * gate[:, None, :]).reshape(B * Cout, C).contiguous()).reshape(B, Cout, wrow).float()
    # hi planes agree except where round-off moves a value across a half boundary; compare the VALUES hi + lo instead
    def value(t):
        t = t.reshape(B, Cout, wrow // 64, 2, 32)
        return t[:, :, :, 0] + t[:, :, :, 1]
    err = (value(wb.float()) - value(want)).abs()
    assert float(err.max()) <= 4e-6 * float(value(want).abs().max()), float(err.max())
    assert float((value(wb.float()) - value(wb_old.float())).abs().max()) <= 4e-6 * float(value(want).abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
def test_se_fold_equals_gated_activation(dtype):
    """(x * gate) @ W^T == x @ (W * gate)^T with the gate computed from the squeezed means."""
    B, HW, C, R, Cout = 3, 300, 448, 28, 112
    x = q(rnd(B, HW, C, seed=1), dtype)
    wr, br = rnd(R, C, seed=2, scale=0.05), rnd(R, seed=3, scale=0.1)
    we, be = rnd(C, R, seed=4, scale=0.2), rnd(C, seed=5, scale=0.1)
    wp = q(rnd(Cout, C, seed=6, scale=1.0 / math.sqrt(C)), dtype)
    mean = x.mean(1)
    hid = F.silu(mean @ wr.t() + br)
    gate = torch.sigmoid(hid @ we.t() + be)
    ref = torch.einsum("bmc,nc->bmn", x * gate[:, None, :], wp)
    part = x.sum(1).reshape(B, 1, C).contiguous().to(DEV)
    hidden = torch.empty(B, R, device=DEV)
    ops.se_hidden(part, 1, 1.0 / HW, wr.to(DEV), br.to(DEV), hidden, B, C, R)
    wb = torch.empty(B, Cout, C, dtype=dtype, device=DEV)
    ops.se_fold(wp.to(dtype).to(DEV), wb, hidden, we.t().contiguous().to(DEV), be.to(DEV), B, Cout, C, R)
    wb2 = torch.empty_like(wb)
    ops.se_gate_fold(part, 1, 1.0 / HW, wr.to(DEV), br.to(DEV), we.t().contiguous().to(DEV), be.to(DEV), wp.to(dtype).to(DEV), wb2,
                     B, Cout, C, R)
    torch.cuda.synchronize()
    assert float((wb2.float() - wb.float()).abs().max()) <= {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}.get(dtype, 1e-5) * float(wb.float().abs().max())
    out = ops.new_act(B * HW, Cout, dtype, DEV)
    ops.conv2d(to_act(x.reshape(B * HW, C), dtype), wb2, None, None, out, B, 1, HW, 1, 1, 1, 0, 0, 1, HW, hip.ACT_NONE, None, None,
               per_image_weights=True)
    torch.cuda.synchronize()
    close(out.torch().float().cpu().reshape(B, HW, Cout), ref, dtype, "se fold")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(1, 30, 40, 128, 7), (2, 20, 37, 64, 15), (1, 40, 50, 32, 31), (1, 9, 9, 8, 31), (1, 21, 19, 16, 5),
                                  (2, 17, 33, 8, 13)])
def test_dwconv_large(case, dtype):
    B, H, W, Cc, k = case
    x = q(rnd(B, Cc, H, W, seed=1), dtype)
    w = rnd(Cc, 1, k, k, seed=2, scale=1.0 / k)
    scale, shift = rnd(Cc, seed=3).abs() + 0.5, rnd(Cc, seed=4)
    ref = F.relu(F.conv2d(x, w, None, 1, (k - 1) // 2, 1, Cc) * scale[None, :, None, None] + shift[None, :, None, None])
    wa = w[:, 0].transpose(1, 2).reshape(Cc, k * k).contiguous().to(DEV)   # [C][kx][ky] f32
    out = ops.new_act(B * H * W, Cc, dtype, DEV)
    ops.dwconv_large(to_act(nhwc(x), dtype), wa, scale.to(DEV), shift.to(DEV), out, B, H, W, k, hip.ACT_RELU)
    close(from_nhwc(out.torch(), B, H, W), ref, dtype, f"dwlarge {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 300, 1392, 7), (1, 1200, 128, 16), (3, 77, 672, 1), (1, 5000, 128, 64)])
def test_channel_sum_and_se(case, dtype):
    B, HW, Cc, ns = case
    x = q(rnd(B, HW, Cc, seed=1), dtype)
    part = torch.zeros(B, ns, Cc, device=DEV)
    ops.channel_sum(to_act(x.reshape(B * HW, Cc), dtype), part, B, HW, ns)
    ref = x.sum(1)
    assert torch.allclose(part.sum(1).cpu(), ref, rtol=1e-4, atol=1e-3 * math.sqrt(HW))
    R = 34
    wr, br, we, be = rnd(R, Cc, seed=2, scale=0.05), rnd(R, seed=3), rnd(Cc, R, seed=4, scale=0.2), rnd(Cc, seed=5)
    hid = torch.empty(B, R, device=DEV)
    ops.se_hidden(part, ns, 1.0 / HW, wr.to(DEV), br.to(DEV), hid, B, Cc, R)
    href = F.silu((ref / HW) @ wr.t() + br)
    assert torch.allclose(hid.cpu(), href, rtol=1e-4, atol=1e-5)
    gref = torch.sigmoid(href @ we.t() + be)
    xa = to_act(x.reshape(B * HW, Cc), dtype)
    ops.se_scale(xa, hid, we.t().contiguous().to(DEV), be.to(DEV), B, HW, R)
    close(xa.torch().float().cpu().reshape(B, HW, Cc), x * gref[:, None, :], dtype)
    xb = to_act(x.reshape(B * HW, Cc), dtype)
    ops.scale_channels(xb, gref.to(DEV), B, HW)
    close(xb.torch().float().cpu().reshape(B, HW, Cc), x * gref[:, None, :], dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Cc", [32, 64, 128])
@pytest.mark.parametrize("rows", [1, 257, 4800])
def test_layernorm(rows, Cc, dtype):
    x = q(rnd(rows, Cc, seed=1, scale=3.0) + 0.7, dtype)
    g, b = rnd(Cc, seed=2).abs() + 0.5, rnd(Cc, seed=3)
    res = q(rnd(rows, Cc, seed=4), dtype)
    out = ops.new_act(rows, Cc, dtype, DEV, ld=2 * Cc, zero=True)
    out = ops.Act(out.buf, Cc, Cc)
    ops.layernorm(to_act(x, dtype), g.to(DEV), b.to(DEV), 1e-5, out, rows, to_act(res, dtype))
    close(out.torch().float().cpu(), F.layer_norm(x, (Cc,), g, b, 1e-5) + res, dtype)
    ops.layernorm(to_act(x, dtype), g.to(DEV), b.to(DEV), 1e-6, out, rows, None)
    close(out.torch().float().cpu(), F.layer_norm(x, (Cc,), g, b, 1e-6), dtype)


def lin_attn_ref(qh, kh, vh, S, eps=1e-6):
    """[N,L,h,d], [N,S',h,d] -> attention.py:31-49 with an explicit v_length S."""
    Q, K = F.elu(qh) + 1, F.elu(kh) + 1
    KV = torch.einsum("nshd,nshv->nhdv", K, vh / S)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + eps)
    return torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("heads,d", [(4, 32), (4, 16), (4, 8), (8, 16), (8, 8), (8, 4)])
def test_attention_zone_mode(heads, d, dtype):
    """hist2image: each zone's queries (p1 x p2 pixels) attend to that zone's 16 ToF samples."""
    B, zn, p1, p2, Dm = 2, 3, 2, 5, heads * d
    Z = zn * zn
    qx = q(rnd(B, zn * p1, zn * p2, Dm, seed=1), dtype)
    kx, vx = q(rnd(B * Z, 16, Dm, seed=2), dtype), q(rnd(B * Z, 16, Dm, seed=3), dtype)
    kv = torch.empty(B * Z * heads * d * d, device=DEV)
    ks = torch.empty(B * Z * heads * d, device=DEV)
    n = ops.attn_kv_ws_floats(B * Z, 1, 16, 1, 16, heads, d)
    ws = torch.empty(max(n, 1), device=DEV)
    ops.attn_kv_reduce(to_act(kx.reshape(-1, Dm), dtype), to_act(vx.reshape(-1, Dm), dtype), kv, ks, ws, B * Z, 1, 16, 1, 16,
                       (0, 1, 0, 16), False, 16.0, heads, d)
    out = ops.new_act(B * zn * p1 * zn * p2, Dm, dtype, DEV)
    ops.attn_apply(to_act(qx.reshape(-1, Dm), dtype), kv, ks, out, B, zn * p1, zn * p2, p1, p2, (0, 0, 0, 0), 16.0, heads, d)
    qz = qx.reshape(B, zn, p1, zn, p2, Dm).permute(0, 1, 3, 2, 4, 5).reshape(B * Z, p1 * p2, heads, d)
    ref = lin_attn_ref(qz, kx.reshape(B * Z, 16, heads, d), vx.reshape(B * Z, 16, heads, d), 16)
    ref = ref.reshape(B, zn, zn, p1, p2, Dm).permute(0, 1, 3, 2, 4, 5).reshape(B, zn * p1, zn * p2, Dm)
    close(out.torch().float().cpu().reshape(B, zn * p1, zn * p2, Dm), ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,W,ws,heads,d", [(30, 40, 6, 8, 16), (13, 20, 9, 8, 8), (24, 31, 12, 8, 4)])
def test_attention_window_mode_with_padding(H, W, ws, heads, d, dtype):
    """LSA: zero-padded windows; padded tokens count with K = 1, V = 0 (transformer.py:101-107)."""
    B, Dm = 2, heads * d
    x = q(rnd(B, H, W, 3 * Dm, seed=1), dtype)
    xa = to_act(x.reshape(-1, 3 * Dm), dtype)
    G = B * math.ceil(H / ws) * math.ceil(W / ws)
    kv, ks = torch.empty(G * heads * d * d, device=DEV), torch.empty(G * heads * d, device=DEV)
    ws_buf = torch.empty(max(ops.attn_kv_ws_floats(B, H, W, ws, ws, heads, d), 1), device=DEV)
    ops.attn_kv_reduce(xa.slice(Dm, Dm), xa.slice(2 * Dm, Dm), kv, ks, ws_buf, B, H, W, ws, ws, (0, H, 0, W), True,
                       float(ws * ws), heads, d)
    out = ops.new_act(B * H * W, Dm, dtype, DEV)
    ops.attn_apply(xa.slice(0, Dm), kv, ks, out, B, H, W, ws, ws, (0, 0, 0, 0), float(ws * ws), heads, d)
    pb, pr = (ws - H % ws) % ws, (ws - W % ws) % ws
    xp = F.pad(x, (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, W + pr
    nh, nw = Hp // ws, Wp // ws
    win = xp.reshape(B, nh, ws, nw, ws, 3 * Dm).permute(0, 1, 3, 2, 4, 5).reshape(B * nh * nw, ws * ws, 3, heads, d)
    ref = lin_attn_ref(win[:, :, 0], win[:, :, 1], win[:, :, 2], ws * ws)
    ref = ref.reshape(B, nh, nw, ws, ws, Dm).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, Dm)[:, :H, :W]
    close(out.torch().float().cpu().reshape(B, H, W, Dm), ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("heads,d,H,W", [(4, 8, 120, 160), (4, 32, 30, 40)])
def test_attention_inside_outside_and_global(heads, d, H, W, dtype):
    """DAPM: keys = inside rectangle (split over many waves), queries = outside tokens only."""
    B, Dm = 2, heads * d
    y0, y1, x0, x1 = H // 10, H - H // 12, W // 7, W - W // 9
    x = q(rnd(B, H, W, 3 * Dm, seed=1), dtype)
    xa = to_act(x.reshape(-1, 3 * Dm), dtype)
    S = (y1 - y0) * (x1 - x0)
    kv, ks = torch.empty(B * heads * d * d, device=DEV), torch.empty(B * heads * d, device=DEV)
    ws_buf = torch.empty(max(ops.attn_kv_ws_floats(B, H, W, H, W, heads, d), 1), device=DEV)
    ops.attn_kv_reduce(xa.slice(Dm, Dm), xa.slice(2 * Dm, Dm), kv, ks, ws_buf, B, H, W, H, W, (y0, y1, x0, x1), False, float(S), heads, d)
    out = ops.new_act(B * H * W, Dm, dtype, DEV)
    ops.attn_apply(xa.slice(0, Dm), kv, ks, out, B, H, W, H, W, (y0, y1, x0, x1), float(S), heads, d)
    ins = x[:, y0:y1, x0:x1].reshape(B, S, 3, heads, d)
    ref = lin_attn_ref(x.reshape(B, H * W, 3, heads, d)[:, :, 0], ins[:, :, 1], ins[:, :, 2], S).reshape(B, H, W, Dm).clone()
    ref[:, y0:y1, x0:x1] = 0
    close(out.torch().float().cpu().reshape(B, H, W, Dm), ref, dtype)


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("D,heads,NB,Hq,Wq,qth,qtw", [(128, 4, 2, 16, 16, 4, 4), (128, 8, 1, 30, 40, 6, 6), (64, 4, 2, 14, 14, 7, 7),
                                                     (64, 8, 1, 13, 17, 13, 17), (32, 8, 2, 24, 30, 12, 12), (32, 4, 1, 9, 11, 3, 3)])
def test_loftr_tail_fused_vs_reference(D, heads, NB, Hq, Wq, qth, qtw, dtype):
    """apply -> merge -> norm1 -> mlp -> norm2 -> +x in one kernel vs the same chain in PyTorch (with
    the storage-type rounding points of the unfused path) and vs the unfused HIP kernels."""
    d = D // heads
    rows = NB * Hq * Wq
    ggy, ggx = -(-Hq // qth), -(-Wq // qtw)
    G = NB * ggy * ggx
    S = float(qth * qtw)
    qf = q(rnd(rows, D, seed=1), dtype)
    xf = q(rnd(rows, D, seed=2), dtype)
    kvf = rnd(G, heads, d, d, seed=3, scale=0.3)
    ksf = rnd(G, heads, d, seed=4).abs() + 0.5
    wm = q(rnd(D, D, seed=5, scale=1.0 / math.sqrt(D)), dtype)
    w0 = q(rnd(2 * D, 2 * D, seed=6, scale=1.0 / math.sqrt(2 * D)), dtype)
    w2 = q(rnd(D, 2 * D, seed=7, scale=1.0 / math.sqrt(2 * D)), dtype)
    g1, b1, g2, b2 = rnd(D, seed=8).abs() + 0.5, rnd(D, seed=9), rnd(D, seed=10).abs() + 0.5, rnd(D, seed=11)
    # reference
    tok = torch.arange(rows)
    xq, yq, bq = tok % Wq, (tok // Wq) % Hq, tok // (Wq * Hq)
    gidx = (bq * ggy + yq // qth) * ggx + xq // qtw
    Q = F.elu(qf).add(1).reshape(rows, heads, d)
    num = torch.einsum("rhi,rhij->rhj", Q, kvf[gidx])
    den = torch.einsum("rhi,rhi->rh", Q, ksf[gidx]) + 1e-6
    msg = q((num / den[..., None] * S).reshape(rows, D), dtype)
    y1 = q(F.layer_norm(q(msg @ wm.t(), dtype), (D,), g1, b1, 1e-5), dtype)
    hmid = q(F.relu(torch.cat([xf, y1], 1) @ w0.t()), dtype)
    ref = F.layer_norm(q(hmid @ w2.t(), dtype), (D,), g2, b2, 1e-5) + xf
    # fused kernel
    qa, xa = to_act(qf, dtype, ld=3 * D), to_act(xf, dtype, ld=2 * D)
    out = ops.new_act(rows, D, dtype, DEV, ld=2 * D, zero=True)
    out = ops.Act(out.buf, D, D)
    kvd, ksd = kvf.contiguous().to(DEV), ksf.contiguous().to(DEV)
    wmd, w0d, w2d = (t.to(dtype).to(DEV) for t in (wm, w0, w2))
    ln1, ln2 = (g1.to(DEV), b1.to(DEV)), (g2.to(DEV), b2.to(DEV))
    ops.loftr_tail(qa, kvd, ksd, xa, out, None, wmd, w0d, w2d, ln1, ln2, NB, Hq, Wq, qth, qtw, S, heads)
    torch.cuda.synchronize()
    got = out.torch().float().cpu()
    close(got, ref, dtype, f"loftr tail D={D} heads={heads}")
    assert float(out.buf[:, :D].abs().max()) == 0
    # the kernel projecting q itself (q = x @ wq^T, transformer.py:45) == the same kernel fed the stored projection
    wq = q(rnd(D, D, seed=12, scale=1.0 / math.sqrt(D)), dtype)
    wqd = wq.to(dtype).to(DEV)
    qp = ops.new_act(rows, D, dtype, DEV)
    ops.linear(xa, wqd, None, None, qp, rows)
    o_a, o_b = ops.new_act(rows, D, dtype, DEV), ops.new_act(rows, D, dtype, DEV)
    ops.loftr_tail(qp, kvd, ksd, xa, o_a, None, wmd, w0d, w2d, ln1, ln2, NB, Hq, Wq, qth, qtw, S, heads)
    ops.loftr_tail(None, kvd, ksd, xa, o_b, wqd, wmd, w0d, w2d, ln1, ln2, NB, Hq, Wq, qth, qtw, S, heads)
    torch.cuda.synchronize()
    close(o_b.torch().float().cpu(), o_a.torch().float().cpu(), dtype, f"loftr tail own q D={D} heads={heads}")
    # unfused HIP chain
    msg_a = ops.new_act(rows, D, dtype, DEV)
    ops.attn_apply(qa, kvd, ksd, msg_a, NB, Hq, Wq, qth, qtw, (0, 0, 0, 0), S, heads, d)
    xb = ops.new_act(rows, 2 * D, dtype, DEV)
    xb.buf[:, :D] = xf.to(dtype).to(DEV)
    ops.linear(msg_a, wmd, None, None, xb.slice(D, D), rows, hip.ACT_NONE, None, None, ln=(ln1[0], ln1[1], 1e-5))
    hid = ops.new_act(rows, 2 * D, dtype, DEV)
    ops.linear(xb, w0d, None, None, hid, rows, hip.ACT_RELU)
    out2 = ops.new_act(rows, D, dtype, DEV)
    ops.linear(hid, w2d, None, None, out2, rows, hip.ACT_NONE, xb.slice(0, D), None, ln=(ln2[0], ln2[1], 1e-5))
    torch.cuda.synchronize()
    close(got, out2.torch().float().cpu(), dtype, f"loftr tail vs unfused D={D} heads={heads}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_resize_matches_interpolate(dtype):
    B, Cc = 2, 64
    x = q(rnd(B, Cc, 15, 20, seed=1), dtype)
    dst = ops.new_act(B * 30 * 40, Cc, dtype, DEV, ld=Cc + 40, zero=True)
    ops.resize_bilinear(to_act(nhwc(x), dtype), 15, 20, (0, 0, 15, 20), dst, 30, 40, (0, 0, 30, 40), B)
    close(from_nhwc(dst.torch(), B, 30, 40), F.interpolate(x, size=[30, 40], mode="bilinear", align_corners=True), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_zone_crop_resize_and_masked_scatter(dtype):
    """fusion.py:136-157: crop (with overhang) -> 28->32 bilinear -> ... -> zero invalid zones ->
    32->28 bilinear -> add onto the in-image part of the rectangle."""
    B, Cc, H, W = 2, 32, 30, 40
    sy, sx, tz, zn, p = -3, 6, 28, 8, 4
    x = q(rnd(B, Cc, H, W, seed=1), dtype)
    grid = ops.new_act(B * 32 * 32, Cc, dtype, DEV)
    ops.resize_bilinear(to_act(nhwc(x), dtype), H, W, (sy, sx, tz, tz), grid, 32, 32, (0, 0, 32, 32), B)
    padded = F.pad(x, (0, 0, 3, 3))
    crop = padded[:, :, sy + 3:sy + 3 + tz, sx:sx + tz]
    ref_grid = F.interpolate(crop, size=[32, 32], mode="bilinear", align_corners=True)
    close(from_nhwc(grid.torch(), B, 32, 32), ref_grid, dtype, "crop+resize")
    valid = (torch.rand(B, zn * zn, generator=torch.Generator().manual_seed(3)) > 0.3)
    zf = q(rnd(B, Cc, 32, 32, seed=5), dtype)
    tok = to_act(nhwc(x), dtype)
    ops.resize_bilinear(to_act(nhwc(zf), dtype), 32, 32, (0, 0, 32, 32), tok, H, W, (sy, sx, tz, tz), B,
                        zone_valid=valid.to(torch.uint8).to(DEV), zn=zn, p1=p, p2=p, accumulate=True)
    m = valid.reshape(B, 1, zn, 1, zn, 1).expand(B, 1, zn, p, zn, p).reshape(B, 1, 32, 32).float()
    back = F.interpolate(zf * m, size=[tz, tz], mode="bilinear", align_corners=True)
    ref = x.clone()
    ref[:, :, 0:sy + tz, sx:sx + tz] += back[:, :, -sy:, :]
    close(from_nhwc(tok.torch(), B, H, W), ref, dtype, "masked scatter-add")


@pytest.mark.parametrize("dtype", DTYPES)
def test_rowtable_copy_and_layout(dtype):
    B, H, W, Cc, Hm, Wm = 2, 26, 34, 128, 30, 40
    x = q(rnd(B * H * W, Cc, seed=1), dtype)
    table = rnd(Hm * Wm, Cc, seed=2)
    out = ops.new_act(B * H * W, Cc, dtype, DEV, ld=2 * Cc)
    ops.add_rowtable(to_act(x, dtype), table.to(DEV), out, B * H * W, H, W, Wm, 3, 5)
    ref = x.reshape(B, H, W, Cc) + table.reshape(Hm, Wm, Cc)[3:3 + H, 5:5 + W]
    close(out.torch().float().cpu().reshape(B, H, W, Cc), ref, dtype)
    cp = ops.new_act(B * H * W, Cc, dtype, DEV)
    ops.copy_rows(out, cp, B * H * W)
    assert torch.equal(cp.torch(), out.torch())
    rgb = rnd(2, 3, 10, 12, seed=4)
    o8 = ops.new_act(2 * 120, 8, dtype, DEV)
    ops.rgb_to_nhwc8(rgb.to(DEV), o8, 2, 10, 12)
    got = o8.torch().float().cpu().reshape(2, 10, 12, 8)
    assert torch.equal(got[..., :3], rgb.to(dtype).float().permute(0, 2, 3, 1)) and float(got[..., 3:].abs().max()) == 0
    s = rnd(77, seed=6)
    o8 = ops.new_act(77, 8, dtype, DEV)
    ops.scalar_to_rows8(s.to(DEV), o8, 77)
    got = o8.torch().float().cpu()
    assert torch.equal(got[:, 0], s.to(dtype).float()) and float(got[:, 1:].abs().max()) == 0


@pytest.mark.parametrize("norm", [0, 1, 2])
def test_bin_regressor(norm):
    B, Cc, hid, nb, HW, ns = 3, 128, 256, 256, 1000, 5
    part = rnd(B, ns, Cc, seed=1, scale=30.0)
    w1x1 = rnd(Cc, Cc, seed=2, scale=0.1)
    w0, b0 = rnd(hid, Cc, seed=3, scale=0.1), rnd(hid, seed=4, scale=0.1)
    w1, b1 = rnd(hid, hid, seed=5, scale=0.08), rnd(hid, seed=6, scale=0.1)
    w2, b2 = rnd(nb, hid, seed=7, scale=0.08), rnd(nb, seed=8, scale=0.1)
    edges, centers = torch.empty(B, nb + 1, device=DEV), torch.empty(B, nb, device=DEV)
    d = lambda t: t.to(DEV)
    dt = lambda t: t.t().contiguous().to(DEV)       # weights go in transposed: [n_in][n_out]
    ops.bin_regressor(d(part), ns, 1.0 / HW, dt(w1x1), dt(w0), d(b0), dt(w1), d(b1), dt(w2), d(b2), 1e-3, 10.0, norm, edges, centers,
                      B, Cc, hid, nb)
    y = (part.sum(1) / HW) @ w1x1.t()
    y = F.leaky_relu(y @ w0.t() + b0, 0.01)
    y = F.leaky_relu(y @ w1.t() + b1, 0.01)
    y = y @ w2.t() + b2
    if norm == 0:
        y = torch.relu(y) + 0.1
        y = y / y.sum(1, keepdim=True)
    elif norm == 1:
        y = torch.softmax(y, 1)
    else:
        y = torch.sigmoid(y)
        y = y / y.sum(1, keepdim=True)
    e = torch.cumsum(F.pad((10.0 - 1e-3) * y, (1, 0), value=1e-3), 1)
    assert torch.allclose(edges.cpu(), e, rtol=1e-5, atol=1e-5)
    assert torch.allclose(centers.cpu(), 0.5 * (e[:, :-1] + e[:, 1:]), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("HW", [64, 1000, 76800 // 16])
def test_bin_softmax(HW, dtype):
    B, nb = 2, 256
    logits = q(rnd(B * HW, nb, seed=1, scale=3.0), dtype)
    centers = torch.sort(torch.rand(B, nb, generator=torch.Generator().manual_seed(2)) * 10, dim=1)[0]
    prob = torch.zeros(B, nb, HW, dtype=dtype, device=DEV)
    pred = torch.empty(B, HW, device=DEV)
    ops.bin_softmax(to_act(logits, dtype), centers.to(DEV), prob, pred, B, HW, nb)
    p = torch.softmax(logits.reshape(B, HW, nb), dim=2)
    close(prob.float().cpu(), p.permute(0, 2, 1), dtype, "prob")
    assert torch.allclose(pred.cpu(), (p * centers[:, None, :]).sum(2), rtol=1e-4, atol=1e-4)
    pred2 = torch.empty(B, HW, device=DEV)
    ops.bin_softmax(to_act(logits, dtype), centers.to(DEV), None, pred2, B, HW, nb)
    assert torch.equal(pred, pred2)


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("HW", [128 * 3, 1000, 4808])
def test_bin_head_fused(HW, dtype):
    """conv_out (1x1, 128 -> 256) + softmax + expectation in one kernel vs the unfused torch chain."""
    B, Cin, nb = 2, 128, 256
    x = q(rnd(B * HW, Cin, seed=1), dtype)
    w = q(rnd(nb, Cin, seed=2, scale=0.25), dtype)
    bias = rnd(nb, seed=3)
    centers = torch.sort(torch.rand(B, nb, generator=torch.Generator().manual_seed(2)) * 10, dim=1)[0]
    prob = torch.zeros(B, nb, HW, dtype=dtype, device=DEV)
    pred = torch.empty(B, HW, device=DEV)
    ops.bin_head_fused(to_act(x, dtype), w.to(dtype).to(DEV), bias.to(DEV), centers.to(DEV), prob, pred, B, HW)
    p = torch.softmax((x @ w.t() + bias).reshape(B, HW, nb), dim=2)
    close(prob.float().cpu(), p.permute(0, 2, 1), dtype, "prob")
    assert torch.allclose(pred.cpu(), (p * centers[:, None, :]).sum(2), rtol=2e-3, atol=2e-3)
    pred2 = torch.empty(B, HW, device=DEV)
    ops.bin_head_fused(to_act(x, dtype), w.to(dtype).to(DEV), bias.to(DEV), centers.to(DEV), None, pred2, B, HW)
    assert torch.equal(pred, pred2)


@pytest.mark.parametrize("D,heads,NB,Hq,Wq,qth,qtw", [(128, 8, 2, 5, 7, 2, 3), (64, 8, 1, 16, 12, 7, 7), (32, 8, 2, 11, 9, 11, 9), (128, 4, 3, 4, 6, 1, 1),
                                                      (64, 4, 1, 9, 30, 3, 10), (32, 4, 1, 70, 3, 5, 3)])
def test_loftr_tail_x3(D, heads, NB, Hq, Wq, qth, qtw):
    """The fused LoFTR tail in the default numerics (float32 tensors, f16x3 GEMMs from pre-split weights): q projection (given or computed),
    apply, merge, norm1, mlp, norm2, + x against the float64 chain; ragged last row tile, key groups of every shape."""
    d = D // heads
    rows = NB * Hq * Wq
    ggy, ggx = -(-Hq // qth), -(-Wq // qtw)
    G = NB * ggy * ggx
    S = float(qth * qtw)
    xf = rnd(rows, D, seed=2)
    wq = rnd(D, D, seed=12, scale=1.0 / math.sqrt(D))
    kvf = rnd(G, heads, d, d, seed=3, scale=0.3)
    ksf = rnd(G, heads, d, seed=4).abs() + 0.5
    wm = rnd(D, D, seed=5, scale=1.0 / math.sqrt(D))
    w0 = rnd(2 * D, 2 * D, seed=6, scale=1.0 / math.sqrt(2 * D))
    w2 = rnd(D, 2 * D, seed=7, scale=1.0 / math.sqrt(2 * D))
    g1, b1, g2, b2 = rnd(D, seed=8).abs() + 0.5, rnd(D, seed=9), rnd(D, seed=10).abs() + 0.5, rnd(D, seed=11)
    X = xf.double()
    qf = X @ wq.double().t()
    tok = torch.arange(rows)
    xq, yq, bq = tok % Wq, (tok // Wq) % Hq, tok // (Wq * Hq)
    gidx = (bq * ggy + yq // qth) * ggx + xq // qtw
    Q = F.elu(qf).add(1).reshape(rows, heads, d)
    num = torch.einsum("rhi,rhij->rhj", Q, kvf.double()[gidx])
    den = torch.einsum("rhi,rhi->rh", Q, ksf.double()[gidx]) + 1e-6
    msg = (num / den[..., None] * S).reshape(rows, D)
    y1 = F.layer_norm(msg @ wm.double().t(), (D,), g1.double(), b1.double(), 1e-5)
    hmid = F.relu(torch.cat([X, y1], 1) @ w0.double().t())
    ref = F.layer_norm(hmid @ w2.double().t(), (D,), g2.double(), b2.double(), 1e-5) + X
    xa = to_act(xf, torch.float32, ld=2 * D)
    kvd, ksd = kvf.contiguous().to(DEV), ksf.contiguous().to(DEV)
    P = lambda t: ops.pack_w_x3(t.contiguous().to(DEV))
    ln1, ln2 = (g1.to(DEV), b1.to(DEV)), (g2.to(DEV), b2.to(DEV))
    for own_q in (True, False):
        out = ops.new_act(rows, D, torch.float32, DEV, ld=2 * D, zero=True)
        out = ops.Act(out.buf, D, D)
        qa = None if own_q else to_act(qf.float(), torch.float32, ld=3 * D)
        ops.loftr_tail(qa, kvd, ksd, xa, out, P(wq) if own_q else None, P(wm), P(w0), P(w2), ln1, ln2, NB, Hq, Wq, qth, qtw, S, heads)
        torch.cuda.synchronize()
        _x3_close(out.torch().cpu(), ref, f"x3 loftr tail D={D} heads={heads} own_q={own_q}", tol=2e-5)
        assert float(out.buf[:, :D].abs().max()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D,heads", [(128, 8), (128, 4), (64, 8), (32, 4)])
def test_fused_tails_give_the_same_bits_with_one_two_and_four_waves_per_workgroup(D, heads, dtype):
    """Round 5: the fused LoFTR / LKPM tails run as 1-, 2- or 4-wave workgroups (16 token rows per wave) by the row count -- narrow workgroups for
    a single image, four waves from 8 192 rows.  A wave's rows never meet another wave's, so the three layouts must agree bit for bit
    (cfp_debug_set keys 35 / 39 force the layout); row counts that leave ragged last workgroups in every layout."""
    lib = hip.load()
    x3 = dtype == torch.float32
    g = torch.Generator().manual_seed(11)
    NB, Hq, Wq, qt = 1, 37, 29, 6
    rows, d = NB * Hq * Wq, D // heads
    G = NB * (-(-Hq // qt)) * (-(-Wq // qt))
    x = ops.new_act(rows, D, dtype, DEV); x.buf.copy_(torch.randn(rows, D, generator=g).to(dtype))
    kv = (torch.randn(G * heads, d, d, generator=g) * 0.3).to(DEV); ks = (torch.rand(G * heads, d, generator=g) + 0.5).to(DEV)
    mk = (lambda n, k: ops.pack_w_x3((torch.randn(n, k, generator=g) / k ** 0.5).to(DEV))) if x3 else (lambda n, k: (torch.randn(n, k, generator=g) / k ** 0.5).to(dtype).to(DEV))
    wq, wm, w0, w2 = mk(D, D), mk(D, D), mk(2 * D, 2 * D), mk(D, 2 * D)
    ln1 = (torch.rand(D, generator=g).to(DEV) + 0.5, torch.randn(D, generator=g).to(DEV), 1e-5)
    ln2 = (torch.rand(D, generator=g).to(DEV) + 0.5, torch.randn(D, generator=g).to(DEV), 1e-5)
    key = 35 if x3 else 39
    got = []
    try:
        for w in (4, 2, 1, 0):
            lib.cfp_debug_set(key, w)
            out = ops.new_act(rows, D, dtype, DEV, zero=True)
            ops.loftr_tail(None, kv, ks, x, out, wq, wm, w0, w2, ln1[:2], ln2[:2], NB, Hq, Wq, qt, qt, float(qt * qt), heads)
            torch.cuda.synchronize()
            got.append(out.buf.clone())
            assert bool(torch.isfinite(got[-1].float()).all())
        if x3 and D in (32, 64, 128):
            t = ops.new_act(rows, D, dtype, DEV); t.buf.copy_(torch.randn(rows, D, generator=g))
            w1, w2l = mk(4 * D, D), mk(D, 4 * D)
            b1, b2 = torch.randn(4 * D, generator=g).to(DEV) * 0.1, torch.randn(D, generator=g).to(DEV) * 0.1
            lk = []
            for w in (4, 2, 1):
                lib.cfp_debug_set(key, w)
                out = ops.new_act(rows, D, dtype, DEV, zero=True)
                ops.lkpm_tail(t, x, out, w1, b1, w2l, b2, ln1[0], ln1[1], rows)
                torch.cuda.synchronize()
                lk.append(out.buf.clone())
            assert torch.equal(lk[0], lk[1]) and torch.equal(lk[0], lk[2])
    finally:
        lib.cfp_debug_set(key, 0)
    for o in got[1:]:
        assert torch.equal(got[0], o), (D, heads, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_fused_tails_are_bit_stable_at_d32(dtype):
    """D = 32 makes every GEMM of the fused tails one or two K-steps long: straight-line code in which the hand-over of the shared weight
    stages between consecutive GEMMs is the whole synchronisation.  Round 4 found the float32 (f16x3) kernel handing a stage over while
    another wave's fragment reads were still in flight (109 of 76 800 rows wrong in 1 run of 20, only at this width).  12 runs on 76 800
    rows (config5's 1/4-scale fusion block) must be bit-identical, LoFTR tail (both q paths) and LKPM tail, both kernels."""
    D, heads, NB, Hq, Wq, qth, qtw = 32, 8, 2, 160, 240, 14, 14
    d, rows = D // heads, NB * Hq * Wq
    G = NB * (-(-Hq // qth)) * (-(-Wq // qtw))
    x3 = dtype == torch.float32
    x = to_act(rnd(rows, D, seed=1), dtype, ld=2 * D)
    kv, ks = (rnd(G, heads, d, d, seed=2) * 0.3).to(DEV), (rnd(G, heads, d, seed=3).abs() + 0.5).to(DEV)
    P = (lambda w: ops.pack_w_x3(w.contiguous().to(DEV))) if x3 else (lambda w: w.to(dtype).to(DEV))
    wq, wm = P(rnd(D, D, seed=4, scale=1 / math.sqrt(D))), P(rnd(D, D, seed=5, scale=1 / math.sqrt(D)))
    w0, w2 = P(rnd(2 * D, 2 * D, seed=6, scale=1 / math.sqrt(2 * D))), P(rnd(D, 2 * D, seed=7, scale=1 / math.sqrt(2 * D)))
    ln1, ln2 = (torch.ones(D, device=DEV), torch.zeros(D, device=DEV)), (torch.ones(D, device=DEV), torch.zeros(D, device=DEV))
    qa = to_act(rnd(rows, D, seed=8), dtype, ld=3 * D)
    w1l, w2l = P(rnd(4 * D, D, seed=9, scale=1 / math.sqrt(D))), P(rnd(D, 4 * D, seed=10, scale=1 / math.sqrt(4 * D)))
    b1l, b2l = rnd(4 * D, seed=11).to(DEV), rnd(D, seed=12).to(DEV)
    refs = [None, None, None]
    for it in range(12):
        outs = []
        for own_q in (True, False):
            out = ops.new_act(rows, D, dtype, DEV)
            ops.loftr_tail(None if own_q else qa, kv, ks, x, out, wq if own_q else None, wm, w0, w2, ln1, ln2, NB, Hq, Wq, qth, qtw, float(qth * qtw), heads)
            outs.append(out.buf)
        out = ops.new_act(rows, D, dtype, DEV)
        ops.lkpm_tail(x, qa, out, w1l, b1l, w2l, b2l, ln1[0], ln1[1], rows)
        outs.append(out.buf)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            if refs[i] is None:
                refs[i] = o.clone()
                assert bool(torch.isfinite(o.float()).all())
            else:
                assert torch.equal(refs[i], o), (it, i, int((refs[i] != o).any(1).sum()))


@pytest.mark.parametrize("D,rows", [(32, 5000), (64, 777), (128, 300), (32, 64), (128, 4097)])
def test_lkpm_tail_x3(D, rows):
    """LKPM's LayerNorm -> pwconv1 -> GELU -> pwconv2 -> + input in one kernel, default numerics (float32 tensors, f16x3 GEMMs, the hidden
    width in four quarters), against the float64 chain with the exact erf GELU."""
    t = rnd(rows, D, seed=1) * 1.5 + 0.2
    xin = rnd(rows, D, seed=2)
    w1, b1 = rnd(4 * D, D, seed=3, scale=1 / math.sqrt(D)), rnd(4 * D, seed=4, scale=0.2)
    w2, b2 = rnd(D, 4 * D, seed=5, scale=1 / math.sqrt(4 * D)), rnd(D, seed=6, scale=0.2)
    g, bt = rnd(D, seed=7).abs() + 0.5, rnd(D, seed=8)
    y = F.layer_norm(t.double(), (D,), g.double(), bt.double(), 1e-6)
    h = F.gelu(y @ w1.double().t() + b1.double())
    ref = h @ w2.double().t() + b2.double() + xin.double()
    out = ops.new_act(rows, D, torch.float32, DEV, ld=2 * D, zero=True)
    out = ops.Act(out.buf, D, D)
    P = lambda w: ops.pack_w_x3(w.contiguous().to(DEV))
    ops.lkpm_tail(to_act(t, torch.float32, ld=D + 8), to_act(xin, torch.float32), out, P(w1), b1.to(DEV), P(w2), b2.to(DEV), g.to(DEV), bt.to(DEV), rows)
    torch.cuda.synchronize()
    _x3_close(out.torch().cpu(), ref, f"x3 lkpm tail D={D} rows={rows}", tol=1e-5)
    assert float(out.buf[:, :D].abs().max()) == 0


@pytest.mark.parametrize("HW", [8 * 8, 30 * 40 + 4, 240 * 320])
def test_bin_head_fused_x3(HW):
    """The fused bin head in the default numerics (float32 tensors, f16x3 matrix math, float32 prob written by the kernel) against the
    float64 chain conv_out -> softmax -> expectation; ragged last tile, a row count that is not a multiple of the 64-pixel store rounds."""
    B, Cin, nb = 2, 128, 256
    x = rnd(B * HW, Cin, seed=1)
    w = rnd(nb, Cin, seed=2, scale=0.25)
    bias = rnd(nb, seed=3)
    centers = torch.sort(torch.rand(B, nb, generator=torch.Generator().manual_seed(2)) * 10, dim=1)[0]
    wx = ops.pack_w_x3(w.contiguous().to(DEV))
    prob = torch.zeros(B, nb, HW, dtype=torch.float32, device=DEV)
    pred = torch.empty(B, HW, device=DEV)
    ops.bin_head_fused(to_act(x, torch.float32), wx, bias.to(DEV), centers.to(DEV), prob, pred, B, HW)
    p = torch.softmax((x.double() @ w.double().t() + bias.double()).reshape(B, HW, nb), dim=2)
    assert float((prob.double().cpu() - p.permute(0, 2, 1)).abs().max()) < 2e-6
    want = (p * centers.double()[:, None, :]).sum(2)
    assert float((pred.double().cpu() - want).abs().max() / want.abs().max()) < 3e-6
    pred2 = torch.empty(B, HW, device=DEV)
    ops.bin_head_fused(to_act(x, torch.float32), wx, bias.to(DEV), centers.to(DEV), None, pred2, B, HW)
    assert torch.equal(pred, pred2)


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("rows,Cin,Cout", [(700, 136, 816), (9600, 56, 224), (70000, 32, 96), (300, 1392, 232), (5000, 64, 64)])
def test_pointwise_two_term_weights(rows, Cin, Cout, dtype):
    """CFP_CONV_W2: weight rows [hi | lo] (ops.pack_w2) walk the K loop twice over the same activations.  Against float64 with the
    UNROUNDED weights the result must be as good as the 16-bit output store allows (inputs are exactly representable), i.e. far
    better than the one-term layer; K tails (K % 64 != 0), split-K shapes and the short-K / many-row class (gen-1 in the plan)."""
    x = q(rnd(rows, Cin, seed=1), dtype)
    w = rnd(Cout, Cin, seed=2, scale=1.0 / math.sqrt(Cin))
    sc, sh = 0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(3)), rnd(Cout, seed=4, scale=0.2)
    ref = (x.double() @ w.double().t()) * sc.double() + sh.double()
    outs = {}
    for two in (False, True):
        wp = (ops.pack_w2(w, dtype) if two else w.to(dtype)).to(DEV).contiguous()
        out = ops.new_act(rows, Cout, torch.float32 if False else dtype, DEV)
        ws_b = ops.conv2d_ws_bytes(rows, Cout, 2 * ((Cin + 63) // 64) * 64, ops.DT[dtype])
        ws = torch.empty(max(ws_b // 4, 1), dtype=torch.float32, device=DEV) if ws_b else None
        ops.linear(to_act(x, dtype), wp, sc.to(DEV), sh.to(DEV), out, rows, ws=ws)
        outs[two] = out.torch().float().cpu().double()
    e1 = float((outs[False] - ref).abs().sum() / ref.abs().sum())
    e2 = float((outs[True] - ref).abs().sum() / ref.abs().sum())
    store = float((ref.float().to(dtype).double() - ref).abs().sum() / ref.abs().sum())     # what the output rounding alone costs
    print(f"{dtype} {rows}x{Cin}->{Cout}: rel-L1 one-term {e1:.2e}, two-term {e2:.2e}, output store alone {store:.2e}")
    assert e2 < 1.15 * store + 1e-6 and e2 < e1


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("R", [32, 1000, 8192 + 5])
def test_hist_encoder_fused(R, dtype):
    """Nine pointwise Conv1d + BatchNorm1d + ReLU layers in one launch (csrc/hist_encoder.hip) against the float64 chain; float32
    arithmetic in every storage mode, so only the final store rounds.  Ragged last workgroup (R % 32 != 0)."""
    widths = [1, 32, 32, 32, 64, 64, 64, 128, 128, 128]
    x = rnd(R, seed=1).abs() * 3
    parts, layout, off, ref, taps = [], [], 0, x.double()[:, None], []
    for l in range(9):
        ci, co = widths[l], widths[l + 1]
        w = rnd(co, ci, seed=10 + l, scale=1.5 / math.sqrt(ci))
        sc = 0.5 + torch.rand(co, generator=torch.Generator().manual_seed(30 + l))
        sh = rnd(co, seed=50 + l, scale=0.3)
        row = []
        for t in (w.reshape(-1), sc, sh):
            row.append(off)
            parts.append(t)
            off += (t.numel() + 3) // 4 * 4
            if t.numel() % 4:
                parts.append(torch.zeros(4 - t.numel() % 4))
        layout.append((row[0], row[1], row[2], ci, co))
        ref = torch.relu((ref @ w.double().t()) * sc.double() + sh.double())
        if l % 3 == 2:
            taps.append(ref)
    outs = [ops.new_act(R, c, dtype, DEV) for c in (32, 64, 128)]
    ops.hist_encoder(x.to(DEV), torch.cat(parts).to(DEV), layout, outs, R)
    torch.cuda.synchronize()
    # with the fusion blocks' positional tables added on the way out (row = sample index % 16)
    pes = [rnd(16, c, seed=70 + c, scale=0.2) for c in (32, 64, 128)]
    outs_pe = [ops.new_act(R, c, dtype, DEV) for c in (32, 64, 128)]
    ops.hist_encoder(x.to(DEV), torch.cat(parts).to(DEV), layout, outs_pe, R, [t.to(DEV) for t in pes], 16)
    for o, t, pe in zip(outs_pe, taps, pes):
        want = (t + pe.double()[torch.arange(R) % 16]).float()
        assert float((o.torch().float().cpu() - want).abs().max() / want.abs().max()) < {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    for o, t in zip(outs, taps):
        got = o.torch().float().cpu()
        if dtype == torch.float32:
            assert torch.allclose(got, t.float(), rtol=2e-5, atol=2e-5)
        else:
            assert torch.equal(got, t.float().to(dtype).float()) or float((got - t.float()).abs().max() / t.abs().max()) < (2e-3 if dtype == torch.float16 else 1.6e-2)
            # float32 arithmetic: all but a few elements are the correctly rounded float64 result
            assert float((got != t.float().to(dtype).float()).float().mean()) < 0.02


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,Cout,ns", [(2, 12, 20, 32, 128, 5), (1, 7, 9, 16, 40, 7), (3, 40, 64, 32, 128, 16)])
def test_conv3x3_output_sum_from_input_sums(B, H, W, C, Cout, ns, dtype):
    """cfp_channel_sum + cfp_conv3x3_mean: the spatial sum of a linear 3x3 convolution (zero padding 1, bias) from nine shifted sums of
    its input, against the convolution itself (float64)."""
    x = q(rnd(B * H * W, C, seed=1) + 0.3, dtype)
    w = rnd(Cout, C, 3, 3, seed=2, scale=0.2)
    bias = rnd(Cout, seed=3)
    ref = F.conv2d(x.double().reshape(B, H, W, C).permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).sum((2, 3))
    xa = to_act(x, dtype, ld=C + (8 if dtype != torch.float32 else 4))
    partial = torch.empty(B * ns * C, device=DEV)
    msum = torch.empty(B, Cout, device=DEV)
    ops.channel_sum(xa, partial, B, H * W, ns)
    ops.conv3x3_mean(partial, ns, xa, w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV), bias.to(DEV), msum, B, H, W, Cout)
    torch.cuda.synchronize()
    assert float((msum.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-5


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("B,H,W,Cin,mid", [(2, 30, 40, 136, 816), (1, 15, 20, 232, 1392), (2, 30, 40, 112, 448), (1, 13, 17, 56, 224), (3, 7, 5, 112, 672)])
def test_mbconv_expand_dw_fused(B, H, W, Cin, mid, dtype):
    """conv_pw 1x1 + BN + SiLU -> conv_dw 3x3 + BN + SiLU + squeeze-excite channel sums in one kernel (csrc/mbconv.hip) against
    the float64 chain with the expanded tensor rounded to the storage type where the unfused path stores it.  Ragged tiles
    (13x17, 7x5), Cin that needs K padding (136, 232, 56), channel groups with a short last chunk (816 = 51 x 16)."""
    plan = ops.mbconv_plan(B, H, W, Cin, mid)
    assert plan is not None
    ntile, kp = plan
    x = q(rnd(B * H * W, Cin, seed=1), dtype)
    wpw = q(rnd(mid, Cin, seed=2, scale=1.0 / math.sqrt(Cin)), dtype)
    wdw = q(rnd(mid, 3, 3, seed=3, scale=0.4), dtype)
    g = torch.Generator().manual_seed(4)
    s1, t1 = 0.7 + 0.6 * torch.rand(mid, generator=g), rnd(mid, seed=5, scale=0.3)
    s2, t2 = 0.7 + 0.6 * torch.rand(mid, generator=g), rnd(mid, seed=6, scale=0.3)
    xi = x.double().reshape(B, H, W, Cin).permute(0, 3, 1, 2)
    m1 = F.silu(F.conv2d(xi, wpw.double()[:, :, None, None]) * s1.double()[None, :, None, None] + t1.double()[None, :, None, None])
    m1 = m1.float().to(dtype).double()                                            # the expanded tensor is a 16-bit tile
    m2 = F.silu(F.conv2d(m1, wdw.double()[:, None], padding=1, groups=mid) * s2.double()[None, :, None, None] + t2.double()[None, :, None, None])
    out = ops.new_act(B * H * W, mid, dtype, DEV)
    out.buf.fill_(float("nan"))
    part = torch.full((B * ntile * mid,), float("nan"), device=DEV)
    ops.mbconv_expand_dw(to_act(x, dtype, ld=Cin + 8), ops.pack_mbconv_pw(wpw, dtype).to(DEV), s1.to(DEV), t1.to(DEV),
                         wdw.reshape(mid, 9).t().contiguous().to(dtype).to(DEV), s2.to(DEV), t2.to(DEV), out, part, B, H, W)
    torch.cuda.synchronize()
    close(from_nhwc(out.torch(), B, H, W), m2.float(), dtype, "mid2")
    sums = part.reshape(B, ntile, mid).sum(1).cpu().double()
    ref = m2.sum((2, 3))
    assert float((sums - ref).abs().max() / ref.abs().max()) < (2e-3 if dtype == torch.float16 else 1.2e-2)


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("rows,D,ld", [(1200 * 2 + 7, 128, 256), (4800, 64, 128), (19200 + 33, 32, 64)])
def test_lkpm_tail_fused(rows, D, ld, dtype):
    """LayerNorm(1e-6) -> pwconv1 -> GELU(erf) -> pwconv2 -> + input in one kernel (cfp_lkpm_tail) against the float64 chain with
    the unfused path's rounding points (normalised input and hidden tensor stored in the 16-bit type, one rounding of the output).
    Ragged last workgroup, inputs living in slices of wider buffers."""
    t = q(rnd(rows, D, seed=1).abs() * 1.5, dtype)                      # post-ReLU input
    xin = q(rnd(rows, D, seed=2), dtype)
    w1 = q(rnd(4 * D, D, seed=3, scale=1.0 / math.sqrt(D)), dtype)
    w2 = q(rnd(D, 4 * D, seed=4, scale=1.0 / math.sqrt(4 * D)), dtype)
    b1, b2 = rnd(4 * D, seed=5, scale=0.2), rnd(D, seed=6, scale=0.2)
    g, bt = 1.0 + 0.2 * rnd(D, seed=7), rnd(D, seed=8, scale=0.1)
    ln = F.layer_norm(t.double(), (D,), g.double(), bt.double(), 1e-6).float().to(dtype).double()
    h = F.gelu(ln @ w1.double().t() + b1.double()).float().to(dtype).double()
    ref = (h @ w2.double().t() + b2.double() + xin.double()).float()
    out = ops.new_act(rows, D, dtype, DEV, ld)
    ops.lkpm_tail(to_act(t, dtype), to_act(xin, dtype, ld=ld), out.slice(0, D) if ld != D else out, w1.to(dtype).to(DEV), b1.to(DEV),
                  w2.to(dtype).to(DEV), b2.to(DEV), g.to(DEV), bt.to(DEV), rows)
    torch.cuda.synchronize()
    close(out.torch().float().cpu(), ref, dtype, "lkpm out")


def _head_ref(x, w3, b3, wo, bo, centers, B, H, W, ram_dtype):
    """conv3x3 (+bias) -> [ram rounded to the storage type or not] -> conv_out -> softmax -> expectation, in float64."""
    xi = x.double().reshape(B, H, W, 128).permute(0, 3, 1, 2)
    ram = F.conv2d(xi, w3.double(), b3.double(), padding=1)
    ram_s = ram.float().to(ram_dtype).double() if ram_dtype is not None else ram
    logits = F.conv2d(ram_s, wo.double()[:, :, None, None], bo.double())
    prob = torch.softmax(logits, dim=1)
    pred = (prob * centers.double()[:, :, None, None]).sum(1)
    return ram, prob.reshape(B, 256, H * W), pred.reshape(B, H * W)


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("B,H,W,ld", [(2, 12, 20, 128), (1, 16, 24, 136), (3, 8, 18, 128), (1, 40, 64, 128)])
@pytest.mark.parametrize("hilo", [(True, True), (False, False), (True, False)])
def test_depth_head_fused(B, H, W, ld, hilo, dtype):
    """depth_head.conv3x3 -> conv_out -> softmax -> expectation as one kernel (csrc/head_fused.hip) against the float64 chain:
    halo rows / columns, tiles that straddle rows and images, a ragged last tile (M % 128 != 0), an input pitch > 128, the ram
    test hook, and the two exactness islands (hi + lo conv_out weights, hi + lo ram)."""
    M = B * H * W
    x = q(rnd(M, 128, seed=1), dtype)
    w3 = rnd(128, 128, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * 128))
    b3 = rnd(128, seed=3, scale=0.5)
    wo = rnd(256, 128, seed=4, scale=0.6)
    bo = rnd(256, seed=5)
    centers = torch.sort(torch.rand(B, 256, generator=torch.Generator().manual_seed(2)) * 10, dim=1)[0]
    w3q = ops.round_taps(w3, dtype)                                             # what the engine packs
    w3p = w3q.permute(0, 2, 3, 1).reshape(128, 9 * 128).to(dtype).to(DEV).contiguous()
    wop = ops.permute_wout(wo, dtype, hilo=hilo[0]).to(DEV)
    prob = torch.zeros(B, 256, H * W, dtype=dtype, device=DEV)
    pred = torch.empty(M, device=DEV)
    ram = ops.new_act(M, 128, dtype, DEV)
    xa = to_act(x, dtype, ld=ld)
    ops.depth_head_fused(xa, w3p, None, b3.to(DEV), wop, bo.to(DEV), centers.to(DEV), prob, pred, B, H, W, ram_out=ram, ram_hilo=hilo[1])
    torch.cuda.synchronize()
    wo_eff = wo if hilo[0] else q(wo, dtype)
    ram_ref, p_ref, pred_ref = _head_ref(x, w3q, b3, wo_eff, bo, centers, B, H, W, None if hilo[1] else dtype)
    close(ram.torch().float().cpu(), nhwc(ram_ref.float()), dtype, "ram")
    # logits carry 1e-3-level noise only when ram is rounded; with both islands the probabilities are f32-accurate up to the 16-bit store
    err = (prob.float().cpu() - p_ref.float()).abs().max()
    assert float(err) < (2e-3 if dtype == torch.float16 else 1.2e-2), float(err)
    # both islands on: what is left is the 16-bit input x itself (exact here) and the 2^-16 (bf16) / 2^-22 (fp16) residual of hi + lo
    t = ({torch.bfloat16: 3e-4, torch.float16: 3e-5}[dtype]) if hilo == (True, True) else ({torch.bfloat16: 2e-2, torch.float16: 3e-3}[dtype])
    assert torch.allclose(pred.cpu().reshape(B, H * W), pred_ref.float(), rtol=t, atol=t), \
        float((pred.cpu().reshape(B, H * W) - pred_ref.float()).abs().max())
    pred2 = torch.empty(M, device=DEV)
    ops.depth_head_fused(xa, w3p, None, b3.to(DEV), wop, bo.to(DEV), centers.to(DEV), None, pred2, B, H, W, ram_hilo=hilo[1])
    assert torch.equal(pred, pred2)


# ---- bit-exact checks of the bf16 fast-path kernels ---------------------------------------------------------------
# Small-integer inputs and weights make every product and every f32 partial sum exact, so the only rounding is the final
# f32 -> bf16 store (round-to-nearest-even): the kernels must reproduce the integer convolution BIT FOR BIT, whatever their
# tiling, K order, LDS swizzle or MFMA operand layout.  This is the parity argument for the kernels the f32 mode never runs.
def _int_tensor(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def _bits(t, dtype):
    return t.to(dtype).view(torch.int16)


INT_CONV_CASES = [(2, 20, 24, 40, 48, 3, 1, (1, 1, 1, 1)), (1, 17, 33, 168, 64, 3, 1, (1, 1, 1, 1)), (2, 9, 11, 8, 40, 3, 2, (0, 0, 1, 1)),
                  (1, 1, 700, 136, 816, 1, 1, (0, 0, 0, 0)), (1, 24, 24, 64, 32, 6, 6, (0, 0, 0, 0))]


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("variant", list(range(22)) + [200 + v for v in range(6)] + ["gen1"])
def test_conv_fast_paths_bit_exact_on_integers(variant, dtype):
    lib = hip.load()
    try:
        if variant == "gen1":
            lib.cfp_debug_set(2, 1)
        else:
            lib.cfp_debug_set(0, int(variant))
        for case in INT_CONV_CASES:
            B, H, W, Cin, Cout, k, s, (pt, pl, pb, pr) = case
            if isinstance(variant, int) and variant >= 200 and not (k == 3 and s == 1):
                continue
            x = _int_tensor((B, Cin, H, W), -3, 3, 1)
            w = _int_tensor((Cout, Cin, k, k), -2, 2, 2)
            Ho, Wo = (H + pt + pb - k) // s + 1, (W + pl + pr - k) // s + 1
            ref = F.conv2d(F.pad(x.double(), (pl, pr, pt, pb)), w.double(), None, s).float()      # exact integers (< 2^24)
            out = ops.new_act(B * Ho * Wo, Cout, dtype, DEV)
            wa = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dtype).to(DEV)
            ops.conv2d(to_act(nhwc(x), dtype), wa, None, None, out, B, H, W, k, k, s, pt, pl, Ho, Wo, hip.ACT_NONE, None, None)
            torch.cuda.synchronize()
            got = out.torch().cpu().reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
            assert torch.equal(got.view(torch.int16), _bits(ref, dtype)), f"variant {variant} case {case}"
    finally:
        lib.cfp_debug_set(0, -1)
        lib.cfp_debug_set(2, 0)


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", [(2, 30, 40, 224, 2, (0, 0, 1, 1)), (1, 15, 20, 1392, 1, (1, 1, 1, 1)), (2, 30, 40, 816, 1, (1, 1, 1, 1)),
                                  (1, 9, 7, 64, 2, (1, 1, 1, 1)), (1, 16, 16, 16, 1, (1, 1, 1, 1))])
def test_dwconv3x3_mfma_bit_exact_on_integers(case, dtype):
    B, H, W, Cc, s, (pt, pl, pb, pr) = case
    x = _int_tensor((B, Cc, H, W), -7, 7, 3)
    w = _int_tensor((Cc, 1, 3, 3), -3, 3, 4)
    Ho, Wo = (H + pt + pb - 3) // s + 1, (W + pl + pr - 3) // s + 1
    ref = F.conv2d(F.pad(x.double(), (pl, pr, pt, pb)), w.double(), None, s, 0, 1, Cc).float()
    out = ops.new_act(B * Ho * Wo, Cc, dtype, DEV)
    wa = w.reshape(Cc, 9).t().contiguous().to(dtype).to(DEV)
    ns = ops.dwconv3x3_strips(B, Ho, Wo, Cc, s, ops.DT[dtype])
    part = torch.empty(B, ns, Cc, device=DEV)
    ops.dwconv3x3_sum(to_act(nhwc(x), dtype), wa, torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV), out, part, B, H, W, s, pt, pl,
                      Ho, Wo, hip.ACT_NONE)
    torch.cuda.synchronize()
    got = out.torch().cpu().reshape(B, Ho, Wo, Cc).permute(0, 3, 1, 2)
    assert torch.equal(got.view(torch.int16), _bits(ref, dtype))
    assert torch.equal(part.sum(1).cpu(), ref.sum((2, 3)))           # channel sums of small integers are exact too


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", [(2, 120, 160, 240, 320, 64, 16, 32), (2, 60, 80, 120, 160, 128, 40, 64), (1, 30, 40, 60, 80, 256, 56, 128),
                                  (2, 15, 20, 30, 40, 256, 136, 256), (1, 13, 17, 26, 34, 64, 24, 16), (3, 7, 9, 21, 19, 128, 8, 40)])
def test_upsample_cat_conv3x3_is_bit_identical_to_resize_then_conv(case, dtype):
    """cfp_upsample_cat_conv3x3 (decoder.py:51-58: interpolate(bilinear, align_corners=True) -> cat -> conv3x3 + BN + LeakyReLU in ONE launch,
    the upsampled tensor and the concatenation never in memory) against cfp_resize_bilinear into the concatenation buffer +
    cfp_conv2d_nhwc: identical bits (same taps, same float32 blend, same rounding point), and both against torch."""
    B, Hs, Ws, H, W, Cup, Cskip, Cout = case
    low = q(rnd(B, Cup, Hs, Ws, seed=31), dtype)
    skip = q(rnd(B, Cskip, H, W, seed=32), dtype)
    Cin = Cup + Cskip
    w = q(rnd(Cout, Cin, 3, 3, seed=33, scale=1.0 / math.sqrt(9 * Cin)), dtype)
    scale, shift = (rnd(Cout, seed=34).abs() + 0.5).to(DEV), rnd(Cout, seed=35, scale=0.1).to(DEV)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(dtype).to(DEV)
    a_low = to_act(nhwc(low), dtype)
    cat = ops.new_act(B * H * W, Cin, dtype, DEV)
    cat.buf[:, Cup:] = nhwc(skip).to(dtype).to(DEV)
    out1, out2 = ops.new_act(B * H * W, Cout, dtype, DEV), ops.new_act(B * H * W, Cout, dtype, DEV)
    ops.upsample_cat_conv3x3(a_low, Hs, Ws, cat.slice(Cup, Cskip), wp, scale, shift, out1, B, H, W, hip.ACT_LRELU)
    ops.resize_bilinear(a_low, Hs, Ws, (0, 0, Hs, Ws), cat.slice(0, Cup), H, W, (0, 0, H, W), B)
    ops.conv2d(cat, wp, scale, shift, out2, B, H, W, 3, 3, 1, 1, 1, H, W, hip.ACT_LRELU)
    torch.cuda.synchronize()
    ref_in = torch.cat([q(F.interpolate(low, size=(H, W), mode="bilinear", align_corners=True), dtype), skip], 1)
    ref = F.leaky_relu(F.conv2d(ref_in, w, None, 1, 1) * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None], 0.01)
    close(from_nhwc(out2.torch(), B, H, W), ref, dtype, f"resize + conv {case}")
    close(from_nhwc(out1.torch(), B, H, W), ref, dtype, f"fused {case}")
    d = (out1.buf.float() - out2.buf.float()).abs()
    v, _ = ops.conv2d_plan(B * H * W, Cout, 9 * Cin, ops.DT[dtype], 0, B, 3, 1)
    if 200 <= v < 300:       # the unfused pair runs the same direct 3x3 kernel (same summation order): identical bits
        assert torch.equal(out1.buf.view(torch.int16), out2.buf.view(torch.int16)), f"{int((d > 0).sum())} elements differ, max {float(d.max()):.3e}"
    else:              # the pair's conv is the implicit GEMM (another K order): equal up to float32 summation order, i.e. one 16-bit ulp
        ulp = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
        assert bool((d <= ulp * out2.buf.float().abs() + 1e-3 * ulp * float(out2.buf.float().abs().max())).all()), float(d.max())


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", [(2, 120, 160, 240, 320, 64, 24, 32, 1), (1, 13, 17, 26, 34, 64, 24, 16, 0), (2, 15, 20, 45, 50, 64, 8, 64, 2),
                                  (1, 9, 11, 20, 33, 64, 64, 40, 3), (1, 30, 40, 60, 80, 64, 16, 24, 7)])
def test_upsample_cat_conv3x3_through_the_halo_kernel(case, dtype):
    """The same fused op with the blend done in the whole-depth halo kernel's loader (up4 of the benched forward): against torch, against
    the direct-kernel form (float32 re-association only), and BIT-identical to the halo kernel run on the materialised
    resize + concatenation with the same tile (same taps, same blend arithmetic, same rounding point, same summation order)."""
    B, Hs, Ws, H, W, Cup, Cskip, Cout, variant = case
    lib = hip.load()
    low = q(rnd(B, Cup, Hs, Ws, seed=31), dtype)
    skip = q(rnd(B, Cskip, H, W, seed=32), dtype)
    Cin = Cup + Cskip
    w = q(rnd(Cout, Cin, 3, 3, seed=33, scale=1.0 / math.sqrt(9 * Cin)), dtype)
    scale, shift = (rnd(Cout, seed=34).abs() + 0.5).to(DEV), rnd(Cout, seed=35, scale=0.1).to(DEV)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(dtype).to(DEV)
    a_low = to_act(nhwc(low), dtype)
    cat = ops.new_act(B * H * W, Cin, dtype, DEV)
    cat.buf[:, Cup:] = nhwc(skip).to(dtype).to(DEV)
    o_halo, o_direct, o_pair = (ops.new_act(B * H * W, Cout, dtype, DEV) for _ in range(3))
    try:
        lib.cfp_debug_set(14, 2)
        lib.cfp_debug_set(0, 300 + variant)           # the tile of the fused launch and of the reference launch below
        ops.upsample_cat_conv3x3(a_low, Hs, Ws, cat.slice(Cup, Cskip), wp, scale, shift, o_halo, B, H, W, hip.ACT_LRELU)
        ops.resize_bilinear(a_low, Hs, Ws, (0, 0, Hs, Ws), cat.slice(0, Cup), H, W, (0, 0, H, W), B)
        ops.conv2d(cat, wp, scale, shift, o_pair, B, H, W, 3, 3, 1, 1, 1, H, W, hip.ACT_LRELU)
        lib.cfp_debug_set(0, -1)
        lib.cfp_debug_set(14, 0)
        ops.upsample_cat_conv3x3(a_low, Hs, Ws, cat.slice(Cup, Cskip), wp, scale, shift, o_direct, B, H, W, hip.ACT_LRELU)
        torch.cuda.synchronize()
    finally:
        lib.cfp_debug_set(0, -1)
        lib.cfp_debug_set(14, 1)
    ref_in = torch.cat([q(F.interpolate(low, size=(H, W), mode="bilinear", align_corners=True), dtype), skip], 1)
    ref = F.leaky_relu(F.conv2d(ref_in, w, None, 1, 1) * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None], 0.01)
    close(from_nhwc(o_halo.torch(), B, H, W), ref, dtype, f"halo fused {case}")
    assert torch.equal(o_halo.buf.view(torch.int16), o_pair.buf.view(torch.int16)), "fused != resize + halo conv with the same tile"
    d = (o_halo.buf.float() - o_direct.buf.float()).abs()
    ulp = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    assert bool((d <= ulp * o_direct.buf.float().abs() + 1e-3 * ulp * float(o_direct.buf.float().abs().max())).all()), float(d.max())


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", [(8, 30, 40, 816, 1), (8, 15, 20, 1392, 1), (8, 60, 80, 224, 2), (8, 30, 40, 816, 2), (8, 30, 40, 448, 1),
                                  (2, 26, 34, 672, 1), (2, 13, 17, 1392, 1), (2, 40, 60, 208, 1), (1, 20, 30, 1392, 1), (2, 52, 68, 224, 2),
                                  (1, 7, 5, 16, 1), (3, 16, 16, 80, 1), (1, 33, 130, 48, 2)])
def test_dwconv3x3_stream_kernel_equals_the_single_phase_kernel(case, dtype):
    """dw3x3_stream_kernel (round 3: whole input image requested by LDS-DMA at kernel start, consumed in row steps behind counted
    vmcnt waits, results stored beside the next step's compute) against dw3x3_mfma_kernel (load -> compute -> store), same
    arithmetic: outputs BIT-identical on the encoder's shapes at the benched batch, the training / config-5 / smoke shapes
    (row tails of 1 ... 14 pixels, ragged last row range, last channel block partly empty, stride 2 with TF-SAME padding), with
    the tensors embedded in wider buffers (pitch > C); channel sums equal up to the order of the per-range partial sums."""
    B, H, W, Cc, s = case
    lib = hip.load()
    Ho, Wo = -(-H // s), -(-W // s)
    pt, pl = max((Ho - 1) * s + 3 - H, 0) // 2, max((Wo - 1) * s + 3 - W, 0) // 2
    x = q(rnd(B, Cc, H, W, seed=11), dtype)
    w = q(rnd(Cc, 1, 3, 3, seed=12, scale=0.4), dtype)
    scale, shift = (rnd(Cc, seed=13).abs() + 0.5).to(DEV), rnd(Cc, seed=14).to(DEV)
    wa = w.reshape(Cc, 9).t().contiguous().to(dtype).to(DEV)
    xin = to_act(nhwc(x), dtype, ld=Cc + 24, c0=8)
    res = []
    try:
        for old in (1, 0):
            lib.cfp_debug_set(6, old)
            buf = ops.new_act(B * Ho * Wo, Cc, dtype, DEV, ld=Cc + 16, zero=True)
            out = ops.Act(buf.buf, 8, Cc)
            ns = ops.dwconv3x3_strips(B, Ho, Wo, Cc, s, ops.DT[dtype])
            part = torch.full((B, ns, Cc), float("nan"), device=DEV)
            ops.dwconv3x3_sum(xin, wa, scale, shift, out, part, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
            torch.cuda.synchronize()
            res.append((buf.buf.clone(), part.sum(1).cpu(), ns))
    finally:
        lib.cfp_debug_set(6, 2)            # back to the default (the sliding-window kernel)
    (o_old, s_old, _), (o_new, s_new, ns_new) = res
    ref = F.silu(F.conv2d(F.pad(x, (pl, (Wo - 1) * s + 3 - W - pl, pt, (Ho - 1) * s + 3 - H - pt)), w, None, s, 0, 1, Cc)
                 * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None])
    close(from_nhwc(ops.Act(o_new, 8, Cc).torch(), B, Ho, Wo), ref, dtype, f"dw3x3 stream {case}")
    close(from_nhwc(ops.Act(o_old, 8, Cc).torch(), B, Ho, Wo), ref, dtype, f"dw3x3 single-phase {case}")
    if not torch.equal(o_old.view(torch.int16), o_new.view(torch.int16)):
        d = (o_old.float() - o_new.float()).abs().reshape(B, Ho, Wo, -1)
        idx = torch.nonzero(d > 0)
        raise AssertionError(f"stream kernel output differs {case}: {idx.shape[0]} elements, first {idx[:6].tolist()}, max {float(d.max()):.3e}")
    assert float(o_new[:, :8].float().abs().max()) == 0 and float(o_new[:, 8 + Cc:].float().abs().max()) == 0     # nothing outside the slice
    assert torch.isfinite(s_new).all() and torch.allclose(s_new, s_old, rtol=2e-5, atol=2e-5 * float(s_old.abs().max()))


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", [(8, 30, 40, 816, 1), (8, 15, 20, 1392, 1), (8, 60, 80, 224, 2), (8, 30, 40, 816, 2), (8, 30, 40, 448, 1),
                                  (2, 26, 34, 672, 1), (2, 13, 17, 1392, 1), (2, 40, 60, 208, 1), (1, 20, 30, 1392, 1), (2, 52, 68, 224, 2),
                                  (1, 7, 5, 16, 1), (3, 16, 16, 80, 1), (1, 33, 130, 48, 2), (2, 17, 3, 32, 1)])
def test_dwconv3x3_sliding_window_kernel(case, dtype):
    """dw3x3_slide_kernel (round 3: a wave owns 16 output rows of a 16-channel group and slides along x with a register window of three
    input columns; taps paired along y: 6 diagonal-weight MFMAs and 2 LDS fragment reads per output column; one barrier per workgroup)
    against dw3x3_mfma_kernel and torch: the same products in another float32 summation order -> within one ulp of the storage type of
    the round-2 kernel; channel sums, the squeeze-excite partial dot products and the nothing-outside-the-slice property too."""
    B, H, W, Cc, s = case
    lib = hip.load()
    Ho, Wo = -(-H // s), -(-W // s)
    pt, pl = max((Ho - 1) * s + 3 - H, 0) // 2, max((Wo - 1) * s + 3 - W, 0) // 2
    x = q(rnd(B, Cc, H, W, seed=11), dtype)
    w = q(rnd(Cc, 1, 3, 3, seed=12, scale=0.4), dtype)
    scale, shift = (rnd(Cc, seed=13).abs() + 0.5).to(DEV), rnd(Cc, seed=14).to(DEV)
    R = 24
    wr = rnd(R, Cc, seed=15, scale=0.1).to(DEV)
    wa = w.reshape(Cc, 9).t().contiguous().to(dtype).to(DEV)
    xin = to_act(nhwc(x), dtype, ld=Cc + 24, c0=8)
    res = []
    try:
        for mode in (1, 2):
            lib.cfp_debug_set(6, mode)
            buf = ops.new_act(B * Ho * Wo, Cc, dtype, DEV, ld=Cc + 16, zero=True)
            out = ops.Act(buf.buf, 8, Cc)
            ns = ops.dwconv3x3_strips(B, Ho, Wo, Cc, s, ops.DT[dtype])
            part = torch.full((B, ns, Cc), float("nan"), device=DEV)
            ops.dwconv3x3_sum(xin, wa, scale, shift, out, part, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
            K = ops.dwconv3x3_se_parts(B, Ho, Wo, Cc, s, ops.DT[dtype])
            hpart = torch.full((B, K, R), float("nan"), device=DEV)
            buf2 = ops.new_act(B * Ho * Wo, Cc, dtype, DEV)
            ops.dwconv3x3_se(xin, wa, scale, shift, buf2, wr, hpart, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
            torch.cuda.synchronize()
            assert torch.equal(buf2.buf.view(torch.int16), ops.Act(buf.buf, 8, Cc).torch().contiguous().view(torch.int16))
            res.append((buf.buf.clone(), part.sum(1).cpu(), hpart.sum(1).cpu()))
    finally:
        lib.cfp_debug_set(6, 2)
    (o_old, s_old, h_old), (o_new, s_new, h_new) = res
    ref = F.silu(F.conv2d(F.pad(x, (pl, (Wo - 1) * s + 3 - W - pl, pt, (Ho - 1) * s + 3 - H - pt)), w, None, s, 0, 1, Cc)
                 * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None])
    close(from_nhwc(ops.Act(o_new, 8, Cc).torch(), B, Ho, Wo), ref, dtype, f"dw3x3 slide {case}")
    ulp = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    d = (o_old.float() - o_new.float()).abs()
    assert bool((d <= ulp * o_old.float().abs() + 1e-6).all()), f"{int((d > ulp * o_old.float().abs() + 1e-6).sum())} elements beyond one ulp, max {float(d.max()):.3e}"
    assert float((d > 0).float().mean()) < 0.02                    # and nearly all of them identical
    assert float(o_new[:, :8].float().abs().max()) == 0 and float(o_new[:, 8 + Cc:].float().abs().max()) == 0
    assert torch.isfinite(s_new).all() and torch.allclose(s_new, s_old, rtol=1e-4, atol=1e-4 * float(s_old.abs().max()))
    assert torch.isfinite(h_new).all() and torch.allclose(h_new, h_old, rtol=1e-4, atol=1e-4 * float(h_old.abs().max()))


@pytest.mark.parametrize("case", [(8, 30, 40, 816, 1), (8, 15, 20, 1392, 1), (8, 60, 80, 224, 2), (8, 30, 40, 816, 2), (8, 30, 40, 448, 1),
                                  (2, 26, 34, 672, 1), (2, 13, 17, 1392, 1), (2, 40, 60, 208, 1), (2, 52, 68, 224, 2),
                                  (1, 7, 5, 16, 1), (3, 16, 16, 80, 1), (1, 33, 130, 48, 2), (2, 17, 3, 40, 1), (1, 1, 1, 8, 1), (2, 2, 9, 8, 2)])
def test_dw3x3_rows_kernel_float32(case):
    """dw3x3_rows_kernel (round 5; float32 storage = the default f16x3 mode's depthwise 3x3: a wave slides down the columns of 8 pixel
    slots with the 3x3 window in registers, borders through the buffer descriptor's out-of-range zeros, no LDS) against the round-1
    LDS-strip kernel it replaces and torch: the stored tensor is BIT-IDENTICAL to the old kernel's (same tap order, same epilogue),
    at the plan's run length and at forced ones (1 row, 3 rows, the whole height); channel sums equal to float32 round-off; nothing
    is written outside the output's column slice; input and output live in wider buffers."""
    B, H, W, Cc, s = case
    lib = hip.load()
    dtype = torch.float32
    Ho, Wo = -(-H // s), -(-W // s)
    pt, pl = max((Ho - 1) * s + 3 - H, 0) // 2, max((Wo - 1) * s + 3 - W, 0) // 2
    x = rnd(B, Cc, H, W, seed=11)
    w = rnd(Cc, 1, 3, 3, seed=12, scale=0.4)
    scale, shift = (rnd(Cc, seed=13).abs() + 0.5).to(DEV), rnd(Cc, seed=14).to(DEV)
    wa = w.reshape(Cc, 9).t().contiguous().to(DEV)
    xin = to_act(nhwc(x), dtype, ld=Cc + 12, c0=4)
    RD = 24
    wr = rnd(RD, Cc, seed=15, scale=0.1).to(DEV)
    res = []
    try:
        for mode, force_r in ((0, 0), (1, 0), (1, 1), (1, 3), (1, 10 ** 6)):
            lib.cfp_debug_set(10, mode)
            lib.cfp_debug_set(11, force_r)
            buf = ops.new_act(B * Ho * Wo, Cc, dtype, DEV, ld=Cc + 8, zero=True)
            out = ops.Act(buf.buf, 4, Cc)
            ns = ops.dwconv3x3_strips(B, Ho, Wo, Cc, s, ops.DT[dtype])
            part = torch.full((B, ns, Cc), float("nan"), device=DEV)
            ops.dwconv3x3_sum(xin, wa, scale, shift, out, part, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
            buf2 = ops.new_act(B * Ho * Wo, Cc, dtype, DEV)
            ops.dwconv3x3(xin, wa, scale, shift, buf2, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)          # without sums: same tensor
            K = ops.dwconv3x3_se_parts(B, Ho, Wo, Cc, s, ops.DT[dtype])
            hpart = torch.full((B, K, RD), float("nan"), device=DEV)
            buf3 = ops.new_act(B * Ho * Wo, Cc, dtype, DEV)
            ops.dwconv3x3_se(xin, wa, scale, shift, buf3, wr, hpart, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)   # with the reduce FC's partial dot products
            torch.cuda.synchronize()
            assert torch.equal(buf2.buf, out.torch().contiguous()) and torch.equal(buf3.buf, buf2.buf)
            res.append((buf.buf.clone(), part.sum(1).cpu(), ns, hpart.sum(1).cpu()))
    finally:
        lib.cfp_debug_set(10, 1)
        lib.cfp_debug_set(11, 0)
    ref = F.silu(F.conv2d(F.pad(x, (pl, (Wo - 1) * s + 3 - W - pl, pt, (Ho - 1) * s + 3 - H - pt)), w, None, s, 0, 1, Cc)
                 * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None])
    o_old, s_old, _, h_old = res[0]
    close(from_nhwc(ops.Act(o_old, 4, Cc).torch(), B, Ho, Wo), ref, dtype, f"dw3x3 old {case}")
    want_h = ref.sum((2, 3)) @ wr.cpu().t()
    for o_new, s_new, ns, h_new in res[1:]:
        assert torch.isfinite(h_new).all() and torch.allclose(h_new, want_h, rtol=1e-4, atol=1e-4 * float(want_h.abs().max()))
        assert torch.equal(o_new, o_old), f"{int((o_new != o_old).sum())} elements differ from the LDS-strip kernel (slots {ns})"
        assert float(o_new[:, :4].abs().max()) == 0 and float(o_new[:, 4 + Cc:].abs().max()) == 0
        assert torch.isfinite(s_new).all() and torch.allclose(s_new, s_old, rtol=1e-5, atol=1e-5 * float(s_old.abs().max()))
    want = ref.sum((2, 3))
    assert torch.allclose(res[1][1], want, rtol=1e-4, atol=1e-4 * float(want.abs().max()))


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", [(1, 40, 50, 32, 31), (2, 33, 47, 16, 15), (1, 30, 40, 128, 7), (1, 64, 32, 8, 31)])
def test_dwconv_large_toeplitz_bit_exact_on_integers(case, dtype):
    B, H, W, Cc, k = case
    x = _int_tensor((B, Cc, H, W), -3, 3, 5)
    w = _int_tensor((Cc, 1, k, k), -1, 1, 6)
    ref = F.conv2d(x.double(), w.double(), None, 1, (k - 1) // 2, 1, Cc).float()
    out = ops.new_act(B * H * W, Cc, dtype, DEV)
    ops.dwconv_large_mfma(to_act(nhwc(x), dtype), ops.toeplitz_bands(w, dtype).to(DEV), torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV),
                          out, B, H, W, k, hip.ACT_NONE)
    torch.cuda.synchronize()
    got = out.torch().cpu().reshape(B, H, W, Cc).permute(0, 3, 1, 2)
    assert torch.equal(got.view(torch.int16), _bits(ref, dtype))
