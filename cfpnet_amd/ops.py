"""Thin tensor-level wrappers over the C ABI: one Python function per HIP entry point.

Activations are `Act` handles: a channel slice [c0, c0+C) of a row-major [rows, ld] device
buffer (NHWC with an explicit pitch), which is how producers write straight into concatenation
buffers.  torch is used for device memory and the stream only.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import hip

DT = {torch.float32: hip.F32, torch.bfloat16: hip.BF16, torch.float16: hip.F16}


@dataclass
class Act:
    buf: torch.Tensor      # [rows, ld], contiguous, float32 / bfloat16 / float16, on the GPU
    c0: int
    C: int

    @property
    def rows(self) -> int:
        return self.buf.shape[0]

    @property
    def ld(self) -> int:
        return self.buf.shape[1]

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr() + self.c0 * self.buf.element_size()

    @property
    def dt(self) -> int:
        return DT[self.buf.dtype]

    def slice(self, c0: int, C: int) -> "Act":
        assert 0 <= c0 and c0 + C <= self.C
        return Act(self.buf, self.c0 + c0, C)

    def torch(self) -> torch.Tensor:
        return self.buf[:, self.c0:self.c0 + self.C]


def new_act(rows: int, C: int, dtype, device, ld: Optional[int] = None, zero: bool = False) -> Act:
    ld = ld or C
    f = torch.zeros if zero else torch.empty
    return Act(f((rows, ld), dtype=dtype, device=device), 0, C)


def _s():
    return hip.current_stream()


def ticket_ws(slab_bytes: int, device) -> torch.Tensor:
    """Split-K workspace with the ticket area of CFP_CONV_WS_TICKETS in front: [CONV_TICKET_BYTES of zeros | slabs].  The kernels hand the area
    back zeroed after every launch, so it is zeroed exactly once, here."""
    ws = torch.zeros((hip.CONV_TICKET_BYTES + slab_bytes + 3) // 4, dtype=torch.float32, device=device)
    ws.cfp_tickets = True
    return ws


def conv2d_ws_bytes(M: int, Cout: int, K: int, dt: int) -> int:
    return int(hip.load().cfp_conv2d_ws_bytes(M, Cout, K, dt))


PLAN_IN_FLIGHT = False      # set by Engine.plan_mode while it records in-flight slots: conv2d() then passes the CFP_CONV_IN_FLIGHT hint


def conv2d(x: Act, w: torch.Tensor, scale, shift, out: Act, B, H, W, KH, KW, stride, pad_t, pad_l, Ho, Wo,
           act=hip.ACT_NONE, residual: Optional[Act] = None, ws: Optional[torch.Tensor] = None, ln=None,
           per_image_weights: bool = False):
    """`ln = (gamma, beta, eps)` fuses a LayerNorm over the output channels (residual added after it);
    `per_image_weights`: w is [B, Cout, K] and image b uses w[b]; a `ws` made by ticket_ws() starts with the zeroed ticket area
    (CFP_CONV_WS_TICKETS: a split-K layer then finishes in its own launch)."""
    tickets = getattr(ws, "cfp_tickets", False)
    assert x.rows >= B * H * W and out.rows >= B * Ho * Wo
    K = KH * KW * x.C
    wshape = (B, out.C, K) if per_image_weights else (out.C, K)
    x3 = x.buf.dtype == torch.float32 and w.dtype == torch.float16      # pre-split f16x3 operand (pack_w_x3) on float32 tensors
    if x3:
        wshape = wshape[:-1] + ((K + 31) // 32 * 64,)
    w2 = (not x3) and (not per_image_weights) and KH * KW == 1 and tuple(w.shape) == (out.C, 2 * ((K + 63) // 64) * 64)     # two-term rows [hi | lo]: pack_w2
    assert (x3 or w.dtype == x.buf.dtype) and (w2 or tuple(w.shape) == wshape) and w.is_contiguous(), (w.shape, wshape)
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    hip.call("cfp_conv2d_nhwc_ex", x.ptr, x.ld, w.data_ptr(), hip.ptr(scale), hip.ptr(shift),
             residual.ptr if residual else 0, residual.ld if residual else 0, out.ptr, out.ld,
             B, H, W, x.C, out.C, KH, KW, stride, pad_t, pad_l, Ho, Wo, act, x.dt,
             hip.ptr(g), hip.ptr(b), float(eps),
             int(per_image_weights) | (hip.CONV_W2 if w2 else 0) | (hip.CONV_IN_FLIGHT if PLAN_IN_FLIGHT else 0) | (hip.CONV_X3 if x3 else 0) |
             (hip.CONV_WS_TICKETS if tickets and ws is not None else 0),
             hip.ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0, _s())


def conv2d_moments(x: Act, w: torch.Tensor, bias, out: Act, B, H, W, KH, KW, stride, pad_t, pad_l, Ho, Wo, ws: Optional[torch.Tensor] = None):
    """conv2d (no activation / residual) + per-row-tile channel moments of the stored output for a following batch-statistics BatchNorm.
    -> (partials [tiles, 2, Cout] float32, nsplit, rows_per_split); nsplit == 0: the kernel chosen does not produce them."""
    import ctypes
    M = B * Ho * Wo
    assert w.dtype == x.buf.dtype and tuple(w.shape) == (out.C, KH * KW * x.C) and w.is_contiguous()
    mom = torch.empty(((M + 63) // 64) * 2 * out.C, dtype=torch.float32, device=x.buf.device)
    ns, rps = ctypes.c_int(0), ctypes.c_int(0)
    hip.call("cfp_conv2d_nhwc_moments", x.ptr, x.ld, w.data_ptr(), hip.ptr(bias), out.ptr, out.ld, B, H, W, x.C, out.C, KH, KW, stride,
             pad_t, pad_l, Ho, Wo, x.dt, hip.ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0, mom.data_ptr(), mom.numel(),
             ctypes.byref(ns), ctypes.byref(rps), _s())
    return mom, ns.value, rps.value


# tile shape (BM, BN, LDS stages) of second-generation variant v (conv_igemm2.hip kCfg)
GEN2_TILES = [(128, 128, 3), (128, 128, 2), (128, 64, 3), (128, 64, 4), (64, 64, 3), (64, 64, 4), (256, 32, 3), (256, 32, 2),
              (128, 32, 3), (128, 32, 4), (256, 16, 2), (128, 16, 4), (64, 128, 3), (64, 64, 2), (128, 64, 2), (64, 128, 2), (128, 32, 2),
              (32, 64, 3), (32, 128, 3), (64, 64, 2), (64, 64, 3), (64, 128, 2)]      # 19-21: eight waves, two K groups
X3_ONLY_TILES = [(256, 128, 2), (256, 64, 2), (64, 64, 2), (128, 64, 2), (128, 128, 2), (64, 128, 2),      # conv_igemm_x3.hip ids 22-27
                 (128, 128, 2), (128, 64, 2), (64, 64, 2), (256, 64, 2), (128, 32, 2), (256, 128, 2),     # 28-33: A-direct (A values global -> registers)
                 (64, 64, 2), (128, 32, 2), (64, 128, 2)]      # 34-36: ids 13 / 16 / 15 compiled for one more resident workgroup per CU (in-flight plans)
X3_ADIRECT = range(28, 34)
X3_OCC = range(34, 37)
GEN1_TILES = [(256, 16), (256, 32), (128, 64), (128, 128)]
# (tile rows, BN) of direct 3x3 variant v (conv3x3_direct.hip kCfg3); pixel tile = rows x 16
DIRECT3_TILES = [(8, 128), (8, 64), (16, 64), (16, 32), (8, 32), (16, 16)]


def conv2d_kernel_name(variant: int, splits: int, dt: int) -> str:
    t = {hip.BF16: "bf16", hip.F16: "f16"}.get(dt, "f32")
    if variant >= 500:
        n = "conv3x3_halo_x3"
    elif variant >= 400:
        bm, bn, st = (GEN2_TILES + X3_ONLY_TILES)[variant - 400]
        n = f"igemm_x3<{bm}x{bn},s{st}" + (",kg2>" if 19 <= variant - 400 <= 21 else ",ad>" if variant - 400 in X3_ADIRECT else ",occ>" if variant - 400 in X3_OCC else ">")
    elif variant >= 300:
        n = f"conv3x3_halo<{t}>"
    elif variant >= 200:
        th, bn = DIRECT3_TILES[variant - 200]
        n = f"conv3x3_direct<{t},{th}x16px,{bn}>"
    elif variant >= 100:
        bm, bn, st = GEN2_TILES[variant - 100]
        n = f"igemm2<{t},{bm}x{bn},s{st}" + (",kg2>" if variant - 100 >= 19 else ">")
    else:
        bm, bn = GEN1_TILES[variant]
        n = f"conv_igemm<{t},{bm}x{bn}>"
    return n + (f"+splitK" if splits > 1 else "")


def conv2d_plan(M: int, Cout: int, K: int, dt: int, rows_per_batch: int = 0, B: int = 1, KH: int = 1, stride: int = 1):
    import ctypes
    v, s = ctypes.c_int(0), ctypes.c_int(0)
    hip.load().cfp_conv2d_plan(M, Cout, K, KH, stride, dt, rows_per_batch, B, ctypes.byref(v), ctypes.byref(s))
    return v.value, s.value


def linear(x: Act, w: torch.Tensor, scale, shift, out: Act, rows: int, act=hip.ACT_NONE, residual: Optional[Act] = None,
           ws: Optional[torch.Tensor] = None, ln=None):
    conv2d(x, w, scale, shift, out, 1, 1, rows, 1, 1, 1, 0, 0, 1, rows, act, residual, ws, ln)


def dwconv3x3(x: Act, w, scale, shift, out: Act, B, H, W, stride, pad_t, pad_l, Ho, Wo, act):
    hip.call("cfp_dwconv3x3_nhwc", x.ptr, x.ld, w.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.ptr, out.ld,
             B, H, W, x.C, stride, pad_t, pad_l, Ho, Wo, act, x.dt, _s())


def dwconv3x3_strips(B: int, Ho: int, Wo: int, C: int, stride: int, dt: int) -> int:
    return int(hip.load().cfp_dwconv3x3_strips(B, Ho, Wo, C, stride, dt))


def dwconv3x3_sum(x: Act, w, scale, shift, out: Act, partial: torch.Tensor, B, H, W, stride, pad_t, pad_l, Ho, Wo, act):
    assert partial.dtype == torch.float32 and partial.numel() >= B * dwconv3x3_strips(B, Ho, Wo, x.C, stride, x.dt) * x.C
    hip.call("cfp_dwconv3x3_sum_nhwc", x.ptr, x.ld, w.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.ptr, out.ld,
             partial.data_ptr(), B, H, W, x.C, stride, pad_t, pad_l, Ho, Wo, act, x.dt, _s())


def se_fold(w_proj: torch.Tensor, w_out: torch.Tensor, hidden, we_t, be, B, Cout, C, R):
    assert w_proj.shape == (Cout, C) and w_out.shape == (B, Cout, C) and w_proj.dtype == w_out.dtype and we_t.shape == (R, C)
    hip.call("cfp_se_fold", w_proj.data_ptr(), w_out.data_ptr(), hidden.data_ptr(), we_t.data_ptr(), be.data_ptr(),
             B, Cout, C, R, DT[w_proj.dtype], _s())


def se_gate_fold(partial, nsplit, inv_hw, wr, br, we_t, be, w_proj: torch.Tensor, w_out: torch.Tensor, B, Cout, C, R):
    """`w_out` float16 with float32 `w_proj`: the per-image pre-split f16x3 operands [B, Cout, ceil(C/32)*64] (zero-initialised by the caller:
    the K padding is never written)."""
    x3 = w_proj.dtype == torch.float32 and w_out.dtype == torch.float16
    assert w_proj.shape == (Cout, C) and w_out.shape == ((B, Cout, (C + 31) // 32 * 64) if x3 else (B, Cout, C)) and (x3 or w_proj.dtype == w_out.dtype)
    assert we_t.shape == (R, C) and wr.shape == (R, C) and partial.numel() >= B * nsplit * C
    hip.call("cfp_se_gate_fold", partial.data_ptr(), nsplit, float(inv_hw), wr.data_ptr(), br.data_ptr(), we_t.data_ptr(), be.data_ptr(),
             w_proj.data_ptr(), w_out.data_ptr(), B, Cout, C, R, hip.F32X3 if x3 else DT[w_proj.dtype], _s())


def dwconv3x3_se_parts(B: int, Ho: int, Wo: int, C: int, stride: int, dt: int) -> int:
    return int(hip.load().cfp_dwconv3x3_se_parts(B, Ho, Wo, C, stride, dt))


def dwconv3x3_se(x: Act, w, scale, shift, out: Act, w_reduce: torch.Tensor, hpart: torch.Tensor, B, H, W, stride, pad_t, pad_l, Ho, Wo, act):
    """Depthwise 3x3 + BN + activation + the squeeze-excite reduce FC's partial dot products (hpart [B][K][R])."""
    R = w_reduce.shape[0]
    assert w_reduce.dtype == torch.float32 and w_reduce.shape == (R, x.C) and hpart.dtype == torch.float32
    assert hpart.numel() >= B * dwconv3x3_se_parts(B, Ho, Wo, x.C, stride, x.dt) * R
    hip.call("cfp_dwconv3x3_se_nhwc", x.ptr, x.ld, w.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.ptr, out.ld,
             w_reduce.data_ptr(), R, hpart.data_ptr(), B, H, W, x.C, stride, pad_t, pad_l, Ho, Wo, act, x.dt, _s())


def se_gate_fold2(hpart, K, inv_hw, br, we_t, be, w_proj32: torch.Tensor, w_out: torch.Tensor, B, Cout, C, R, x3: bool = False):
    """`x3`: w_out is the per-image pre-split f16x3 operand [B, Cout, ceil(C/32)*64] float16 (zero-initialised by the caller)."""
    assert w_proj32.shape == (Cout, C) and w_proj32.dtype == torch.float32
    assert w_out.shape == ((B, Cout, (C + 31) // 32 * 64) if x3 else (B, Cout, C)) and (not x3 or w_out.dtype == torch.float16)
    assert we_t.shape == (R, C) and hpart.dtype == torch.float32 and hpart.numel() >= B * K * R
    hip.call("cfp_se_gate_fold2", hpart.data_ptr(), K, float(inv_hw), br.data_ptr(), we_t.data_ptr(), be.data_ptr(),
             w_proj32.data_ptr(), w_out.data_ptr(), B, Cout, C, R, hip.F32X3 if x3 else DT[w_out.dtype], _s())


def dwconv_large(x: Act, w, scale, shift, out: Act, B, H, W, k, act):
    hip.call("cfp_dwconv_large_nhwc", x.ptr, x.ld, w.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.ptr, out.ld,
             B, H, W, x.C, k, act, x.dt, _s())


def _bracket16(w: torch.Tensor, dtype):
    """(down, up): the largest value representable in `dtype` that is <= w and the smallest that is >= w, as float32."""
    r16 = w.to(dtype)
    r = r16.float()
    bits = r16.view(torch.int16).to(torch.int32)
    toward_inf = torch.where(r >= 0, bits + 1, bits - 1)          # next representable value away from zero ... for r < 0 towards zero
    toward_ninf = torch.where(r > 0, bits - 1, bits + 1)
    # r == +0: below it is the smallest negative subnormal (0x8001); r == -0 (bits -32768): above it is 0x0001
    toward_ninf = torch.where(r16.view(torch.int16) == 0, torch.full_like(bits, -32767), toward_ninf)
    toward_inf = torch.where(r16.view(torch.int16) == -32768, torch.ones_like(bits), toward_inf)
    nxt_up = toward_inf.to(torch.int16).view(dtype).float()
    nxt_dn = toward_ninf.to(torch.int16).view(dtype).float()
    up = torch.where(r >= w, r, nxt_up)
    down = torch.where(r <= w, r, nxt_dn)
    return down, up


def pack_w_x3(w2d: torch.Tensor) -> torch.Tensor:
    """[rows, K] float32 on the GPU -> the pre-split f16x3 operand [rows, ceil(K/32)*64] float16 (cfp_pack_w_x3): per 32-channel K-step
    [hi(32) | lo(32)] in the kernels' lane order, zero padded."""
    assert w2d.is_cuda and w2d.dtype == torch.float32 and w2d.dim() == 2 and w2d.is_contiguous()
    rows, k = w2d.shape
    out = torch.empty(rows, (k + 31) // 32 * 64, dtype=torch.float16, device=w2d.device)
    hip.call("cfp_pack_w_x3", w2d.data_ptr(), out.data_ptr(), rows, k, _s())
    if os.environ.get("CFP_X3_DIAG_TWO_TERM", "0") == "1":
        # DIAGNOSTIC ONLY (tools/x3_two_term_probe.py, VERDICT r4 task 1b): zero the lo halves of the STATIC weights, i.e. the accuracy of a
        # two-term product A_hi W_hi + A_lo W_hi (weights as ONE half) measured with the three-term kernels.  Never set in product use.
        out.view(rows, -1, 2, 32)[:, :, 1, :] = 0
    return out


def pack_w2(w2d: torch.Tensor, dtype) -> torch.Tensor:
    """Pointwise weights [Cout, K] float32 -> two-term rows [Cout, 2 * Kp] in `dtype` (Kp = K rounded up to 64): hi = round(W),
    lo = round(W - hi), zero padding behind each half (cfp_conv2d_nhwc_ex with CFP_CONV_W2)."""
    w2d = w2d.detach().float()
    co, k = w2d.shape
    kp = (k + 63) // 64 * 64
    hi = w2d.to(dtype)
    lo = (w2d - hi.float()).to(dtype)
    out = torch.zeros(co, 2 * kp, dtype=dtype)
    out[:, :k] = hi
    out[:, kp:kp + k] = lo
    return out.contiguous()


def round_taps(w: torch.Tensor, dtype, bracket: bool = True) -> torch.Tensor:
    """Round the weights of a k x k convolution ([..., kh, kw], float32) to the 16-bit storage format `dtype`, choosing the
    rounding DIRECTION of every weight by error diffusion along the taps: walking the kh x kw window in serpentine order, a weight
    goes to whichever of its two bracketing representable values is nearer to (weight + error left by its predecessors).  Every
    weight stays within one unit in the last place of its float32 value, but the errors of the taps of one (cout, cin) pair cancel
    instead of growing like a random walk, so the part of the layer's rounding error that multiplies the locally constant part of its
    input (feature maps are smooth over a 3 x 3 neighbourhood, and after an activation their mean is not zero) disappears.
    Measured on MI355X (tools/precision_ablation_sd.py, profiles/r2_precision_budget.md): the weight-rounding error of the depth
    head's 3x3 conv drops ~6x in fp16 and ~3x in bf16; no kernel changes, no run-time cost.  Returns float32 holding exactly
    representable values (the cast to `dtype` is then exact)."""
    w = w.detach().float()
    if dtype == torch.float32 or w.shape[-1] * w.shape[-2] == 1:
        return w
    kh, kw = w.shape[-2], w.shape[-1]
    order = [r * kw + (c if r % 2 == 0 else kw - 1 - c) for r in range(kh) for c in range(kw)]
    flat = w.reshape(-1, kh * kw)
    if not bracket:           # experiment only: plain error diffusion may move a small weight by the ulp of a large neighbour
        out = torch.empty_like(flat)
        e = torch.zeros_like(flat[:, 0])
        for t in order:
            v = flat[:, t] + e
            q = v.to(dtype).float()
            e = v - q
            out[:, t] = q
        return out.reshape(w.shape)
    down, up = _bracket16(flat, dtype)
    # the walk along the taps is sequential; numpy (float32, vectorised over all rows) keeps its per-step overhead at microseconds
    import numpy as np
    f, dn, upn = flat.numpy(), down.numpy(), up.numpy()
    out = np.empty_like(f)
    e = np.zeros(f.shape[0], dtype=np.float32)
    for t in order:
        v = f[:, t] + e
        take_down = np.abs(v - dn[:, t]) <= np.abs(upn[:, t] - v)
        q = np.where(take_down, dn[:, t], upn[:, t])
        e = v - q
        out[:, t] = q
    return torch.from_numpy(out).reshape(w.shape)


def toeplitz_bands(w: torch.Tensor, dtype) -> torch.Tensor:
    """[C,1,k,k] or [C,k,k] depthwise weights (ky, kx) -> the banded-Toeplitz B-operand table of cfp_dwconv_large_mfma_nhwc."""
    w = w.detach().float().cpu()
    if w.dim() == 4:
        w = w[:, 0]
    w = round_taps(w, dtype)
    C, k, _ = w.shape
    halo = (k - 1) // 2
    lm = (halo + 7) // 8 * 8
    nh = (16 + lm + halo + 31) // 32
    lane = torch.arange(64)
    h = torch.arange(nh)
    e = torch.arange(8)
    kx = 32 * h[:, None, None] + 8 * (lane[None, :, None] // 16) + e[None, None, :] - (lm - halo) - (lane[None, :, None] % 16)   # [nh,64,8]
    ok = (kx >= 0) & (kx < k)
    g = w[:, :, kx.clamp(0, k - 1)]                      # [C, k(ky), nh, 64, 8]
    g = torch.where(ok[None, None], g, torch.zeros(()))
    assert g.numel() == int(hip.load().cfp_dwconv_large_toeplitz_elems(C, k))
    return g.contiguous().to(dtype)


def toeplitz_bands_x3(w: torch.Tensor) -> torch.Tensor:
    """The band tables of the f16x3 large depthwise kernel: [2, ...] float16 = the table of hi = half(w) and the table of lo = half(w - hi)."""
    w = w.detach().float().cpu()
    if w.dim() == 4:
        w = w[:, 0]
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return torch.stack([toeplitz_bands(hi.float(), torch.float16), toeplitz_bands(lo.float(), torch.float16)]).contiguous()


def toeplitz_bands_dev(w: torch.Tensor, dtype, flip: bool = False) -> torch.Tensor:
    """`toeplitz_bands` for weights that live (and change every step) on the device: [C,k,k] (ky, kx) float32 -> the band table in
    `dtype` (of the 180-degree-rotated kernel when `flip`), one launch (cfp_dwconv_large_toeplitz), capturable in a HIP graph."""
    C, k, _ = w.shape
    assert w.dtype == torch.float32 and w.is_contiguous()
    out = torch.empty(int(hip.load().cfp_dwconv_large_toeplitz_elems(C, k)), dtype=dtype, device=w.device)
    hip.call("cfp_dwconv_large_toeplitz", w.data_ptr(), out.data_ptr(), C, k, int(flip), DT[dtype], _s())
    return out


def dwconv_large_mfma(x: Act, tb: torch.Tensor, scale, shift, out: Act, B, H, W, k, act):
    """16-bit tensors + a 16-bit band table, or float32 tensors + the [hi | lo] float16 tables of toeplitz_bands_x3 (f16x3 matrix math)."""
    x3 = x.buf.dtype == torch.float32 and tb.dtype == torch.float16
    assert not x3 or tb.shape[0] == 2
    hip.call("cfp_dwconv_large_mfma_nhwc", x.ptr, x.ld, tb.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.ptr, out.ld,
             B, H, W, x.C, k, act, hip.F32X3 if x3 else x.dt, _s())


def channel_sum(x: Act, partial: torch.Tensor, B, HW, nsplit):
    assert partial.dtype == torch.float32 and partial.numel() >= B * nsplit * x.C
    hip.call("cfp_channel_sum", x.ptr, x.ld, partial.data_ptr(), B, HW, x.C, nsplit, x.dt, _s())


def se_hidden(partial, nsplit, inv_hw, wr, br, hidden, B, C, R):
    hip.call("cfp_se_hidden", partial.data_ptr(), nsplit, float(inv_hw), wr.data_ptr(), br.data_ptr(), hidden.data_ptr(),
             B, C, R, _s())


def se_scale(x: Act, hidden, we_t, be, B, HW, R):
    assert we_t.shape == (R, x.C)
    hip.call("cfp_se_scale", x.ptr, x.ld, hidden.data_ptr(), we_t.data_ptr(), be.data_ptr(), B, HW, x.C, R, x.dt, _s())


def scale_channels(x: Act, gate, B, HW):
    hip.call("cfp_scale_channels", x.ptr, x.ld, gate.data_ptr(), B, HW, x.C, x.dt, _s())


def layernorm(x: Act, gamma, beta, eps, out: Act, rows, residual: Optional[Act] = None):
    hip.call("cfp_layernorm", x.ptr, x.ld, gamma.data_ptr(), beta.data_ptr(), float(eps),
             residual.ptr if residual else 0, residual.ld if residual else 0, out.ptr, out.ld, rows, x.C, x.dt, _s())


def attn_kv_ws_floats(NB, Hk, Wk, th, tw, heads, d) -> int:
    return int(hip.load().cfp_attn_kv_ws_floats(NB, Hk, Wk, th, tw, heads, d))


def attn_kv_reduce(k: Act, v: Act, kv, ksum, ws, NB, Hk, Wk, th, tw, clip, count_pad, v_length, heads, d):
    cy0, cy1, cx0, cx1 = clip
    hip.call("cfp_attn_kv_reduce", k.ptr, k.ld, v.ptr, v.ld, kv.data_ptr(), ksum.data_ptr(), hip.ptr(ws),
             NB, Hk, Wk, th, tw, cy0, cy1, cx0, cx1, int(count_pad), float(v_length), heads, d, k.dt, _s())


def attn_apply(q: Act, kv, ksum, out: Act, NB, Hq, Wq, qth, qtw, excl, v_length, heads, d, eps=1e-6):
    ey0, ey1, ex0, ex1 = excl
    hip.call("cfp_attn_apply", q.ptr, q.ld, kv.data_ptr(), ksum.data_ptr(), out.ptr, out.ld,
             NB, Hq, Wq, qth, qtw, ey0, ey1, ex0, ex1, float(v_length), float(eps), heads, d, q.dt, _s())


def loftr_tail(q: Optional[Act], kv, ksum, x: Act, out: Act, w_q, w_merge, w_mlp0, w_mlp2, ln1, ln2, NB, Hq, Wq, qth, qtw, v_length,
               heads, eps=1e-6, ln_eps=1e-5):
    """`q` given: the projected queries; `q` None and `w_q` [D, D] given: the kernel projects q = x @ w_q^T for its own rows."""
    D = x.C
    x3 = x.buf.dtype == torch.float32 and w_merge.dtype == torch.float16      # float32 tensors + pre-split f16x3 operands (pack_w_x3)
    assert (q is None) != (w_q is None), "loftr_tail: give either q or w_q"
    k1, k2 = (2 * D, 4 * D) if x3 else (D, 2 * D)                             # packed rows hold 64 halves per 32 channels
    assert out.C == D and w_merge.shape == (D, k1) and w_mlp0.shape == (2 * D, k2) and w_mlp2.shape == (D, k2)
    assert q is None or q.C == D
    assert w_q is None or w_q.shape == (D, k1)
    hip.call("cfp_loftr_tail", q.ptr if q is not None else None, q.ld if q is not None else 0, kv.data_ptr(), ksum.data_ptr(), x.ptr, x.ld,
             out.ptr, out.ld, w_q.data_ptr() if w_q is not None else None, w_merge.data_ptr(),
             w_mlp0.data_ptr(), w_mlp2.data_ptr(), ln1[0].data_ptr(), ln1[1].data_ptr(), ln2[0].data_ptr(), ln2[1].data_ptr(),
             float(ln_eps), NB, Hq, Wq, qth, qtw, float(v_length), float(eps), heads, D, hip.F32X3 if x3 else x.dt, _s())


def resize_bilinear(src: Act, Hs, Ws, srect, dst: Act, Hd, Wd, drect, B, zone_valid=None, zn=0, p1=0, p2=0,
                    accumulate=False):
    sy0, sx0, sh, sw = srect
    dy0, dx0, dh, dw = drect
    assert src.C == dst.C
    hip.call("cfp_resize_bilinear", src.ptr, src.ld, Hs, Ws, sy0, sx0, sh, sw, dst.ptr, dst.ld, Hd, Wd, dy0, dx0, dh, dw,
             hip.ptr(zone_valid), zn, p1, p2, int(accumulate), B, src.C, src.dt, _s())


def add_rowtable(x: Act, table, out: Act, rows, H, W, Wt, oy, ox):
    assert table.dtype == torch.float32 and table.shape[1] == x.C
    hip.call("cfp_add_rowtable", x.ptr, x.ld, table.data_ptr(), out.ptr, out.ld, rows, x.C, H, W, Wt, oy, ox, x.dt, _s())


def copy_rows(x: Act, out: Act, rows):
    hip.call("cfp_copy_rows", x.ptr, x.ld, out.ptr, out.ld, rows, x.C, x.dt, _s())


def upsample_cat_conv3x3(low: Act, Hs, Ws, skip: Act, w, scale, shift, out: Act, B, H, W, act, x3: bool = False):
    """bilinear (align_corners) upsample of `low` to H x W + concat with `skip` + conv3x3 + BN + activation in one launch.  16-bit modes:
    `w` [Cout, 9 * (Cup + Cskip)] in the storage type; `x3` (float32 tensors, f16x3 matrix math): `w` = pack_w_x3_cat's operand."""
    if x3:
        assert low.dt == skip.dt == out.dt == hip.F32 and w.dtype == torch.float16 and w.shape[-1] == 9 * (low.C + (skip.C + 31) // 32 * 32) * 2
        hip.call("cfp_upsample_cat_conv3x3", low.ptr, low.ld, Hs, Ws, low.C, skip.ptr, skip.ld, skip.C, w.data_ptr(),
                 hip.ptr(scale), hip.ptr(shift), out.ptr, out.ld, B, H, W, out.C, act, hip.F32X3, _s())
        return
    assert w.shape[-1] == 9 * (low.C + skip.C) and low.dt == skip.dt == out.dt
    hip.call("cfp_upsample_cat_conv3x3", low.ptr, low.ld, Hs, Ws, low.C, skip.ptr, skip.ld, skip.C, w.data_ptr(),
             hip.ptr(scale), hip.ptr(shift), out.ptr, out.ld, B, H, W, out.C, act, out.dt, _s())


def pack_w_x3_cat(w: torch.Tensor, cup: int) -> torch.Tensor:
    """[Cout, 3, 3, Cup + Cskip] float32 on the GPU (the concatenation's channel order, decoder.py:56-57) -> the f16x3 operand of
    cfp_upsample_cat_conv3x3: the same weights over the PADDED channel axis [Cup | Cskip -> next multiple of 32] (zero weights on the padding;
    Cup % 32 == 0), per tap, through cfp_pack_w_x3."""
    co, kh, kw, cin = w.shape
    assert kh == 3 and kw == 3 and cup % 32 == 0 and 0 < cup < cin
    csk = cin - cup
    pad = (csk + 31) // 32 * 32 - csk
    wp = torch.nn.functional.pad(w, (0, pad)) if pad else w
    return pack_w_x3(wp.reshape(co, 9 * (cup + csk + pad)).contiguous())


def copy_rows2(x0: Act, out0: Act, x1: Act, out1: Act, rows):
    """Two strided row copies in one launch (concatenation / split)."""
    assert x0.dt == x1.dt == out0.dt == out1.dt and out0.C == x0.C and out1.C == x1.C
    hip.call("cfp_copy_rows2", x0.ptr, x0.ld, out0.ptr, out0.ld, x0.C, x1.ptr, x1.ld, out1.ptr, out1.ld, x1.C, rows, x0.dt, _s())


def rgb_to_nhwc8(rgb: torch.Tensor, out: Act, B, H, W):
    assert rgb.dtype == torch.float32 and rgb.is_contiguous() and out.C == 8 and out.ld == 8
    hip.call("cfp_rgb_to_nhwc8", rgb.data_ptr(), out.ptr, B, H, W, out.dt, _s())


def rgb_to_nhwc8_hilo(rgb: torch.Tensor, out: Act, B, H, W):
    """16-bit storage: [hi(3) | lo(3) | 0 0] per pixel -- the stem then sees the float32 image (its weight rows repeat the real channels)."""
    assert rgb.dtype == torch.float32 and rgb.is_contiguous() and out.C == 8 and out.ld == 8
    hip.call("cfp_rgb_to_nhwc8_hilo", rgb.data_ptr(), out.ptr, B, H, W, out.dt, _s())


def scalar_to_rows8(x: torch.Tensor, out: Act, rows):
    assert x.dtype == torch.float32 and x.is_contiguous() and out.C == 8 and out.ld == 8
    hip.call("cfp_scalar_to_rows8", x.data_ptr(), out.ptr, rows, out.dt, _s())


def bin_regressor(partial, nsplit, inv_hw, w1x1, w0, b0, w1, b1, w2, b2, min_val, max_val, norm, edges, centers, B, C,
                  hidden, nbins):
    hip.call("cfp_bin_regressor", partial.data_ptr(), nsplit, float(inv_hw), w1x1.data_ptr(), w0.data_ptr(), b0.data_ptr(),
             w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), float(min_val), float(max_val), norm,
             edges.data_ptr(), centers.data_ptr(), B, C, hidden, nbins, _s())


def bin_softmax(logits: Act, centers, prob, pred, B, HW, nbins):
    hip.call("cfp_bin_softmax", logits.ptr, logits.ld, centers.data_ptr(), hip.ptr(prob), pred.data_ptr(), B, HW, nbins,
             logits.dt, _s())


def bin_head_fused(x: Act, w, bias, centers, prob, pred, B, HW):
    """conv_out (1x1, -> 256 logits) + softmax + expectation in one launch.  16-bit tensors with 16-bit weights [256, C], or float32
    tensors with the pre-split f16x3 operand of pack_w_x3 (float32 prob)."""
    x3 = x.buf.dtype == torch.float32 and w.dtype == torch.float16
    assert not x3 or (tuple(w.shape) == (256, (x.C + 31) // 32 * 64) and (prob is None or prob.dtype == torch.float32))
    hip.call("cfp_bin_head_fused", x.ptr, x.ld, w.data_ptr(), bias.data_ptr(), centers.data_ptr(), hip.ptr(prob),
             pred.data_ptr(), B, HW, x.C, hip.F32X3 if x3 else x.dt, _s())


def permute_wout(w: torch.Tensor, dtype, hilo: bool = True, diffuse: bool = False) -> torch.Tensor:
    """conv_out weights [256, 128(, 1, 1)] float32 -> the operand of cfp_depth_head_fused: input-channel axis in the kernel's
    fragment order, stored as [planes, 256, 128] in `dtype` (plane 0 = round(W), plane 1 = round(W - plane 0) when `hilo`)."""
    w = w.detach().float().reshape(w.shape[0], -1)
    assert w.shape == (256, 128)
    pos = torch.arange(128)
    kb, q, e = pos // 32, (pos % 32) // 8, pos % 8
    src = 32 * kb + 16 * (e // 4) + 4 * q + (e % 4)
    wp = w[:, src]
    hi = (round_taps(w.reshape(256, 1, 128), dtype).reshape(256, 128)[:, src] if diffuse else wp).to(dtype)
    planes = [hi]
    if hilo:
        planes.append((wp - hi.float()).to(dtype))
    return torch.stack(planes).contiguous()


def depth_head_fused(x: Act, w3, scale3, shift3, wout_perm: torch.Tensor, bias_out, centers, prob, pred, B, H, W, ram_out: Optional[Act] = None,
                     ram_hilo: bool = True, probe: int = 0):
    assert x.C == 128 and w3.shape == (128, 9 * 128) and wout_perm.shape[1:] == (256, 128) and wout_perm.dtype == x.buf.dtype
    flags = (hip.HEAD_WOUT_HILO if wout_perm.shape[0] == 2 else 0) | (hip.HEAD_RAM_HILO if ram_hilo else 0) | (probe << 8)
    if ram_out is not None:
        assert ram_out.ld == 128 and ram_out.C == 128
    hip.call("cfp_depth_head_fused", x.ptr, x.ld, w3.data_ptr(), hip.ptr(scale3), hip.ptr(shift3), wout_perm.data_ptr(), bias_out.data_ptr(),
             centers.data_ptr(), hip.ptr(prob), pred.data_ptr(), ram_out.ptr if ram_out is not None else 0, B, H, W, flags, x.dt, _s())


def hist_encoder(hist: torch.Tensor, blob: torch.Tensor, layout, outs, R: int, pe=(None, None, None), n_pe: int = 0):
    """hist [R] f32, blob f32 parameters, layout = list of 9 (w_off, scale_off, shift_off, cin, cout); outs = three Acts [R, cout];
    pe = optional f32 tables [n_pe, cout] added to the taps (row = sample index % n_pe)."""
    import ctypes
    assert hist.dtype == torch.float32 and hist.is_contiguous() and hist.numel() >= R and blob.dtype == torch.float32 and len(layout) == 9
    flat = [int(v) for row in layout for v in row]
    arr = (ctypes.c_int * len(flat))(*flat)
    for o, row in zip(outs, layout[2::3]):
        assert o.C == row[4] and o.ld == o.C and o.c0 == 0 and o.rows >= R
    for t, o in zip(pe, outs):
        assert t is None or (t.dtype == torch.float32 and tuple(t.shape) == (n_pe, o.C) and t.is_contiguous())
    hip.call("cfp_hist_encoder", hist.data_ptr(), blob.data_ptr(), ctypes.addressof(arr), outs[0].ptr, outs[1].ptr, outs[2].ptr,
             hip.ptr(pe[0]), hip.ptr(pe[1]), hip.ptr(pe[2]), n_pe, R, outs[0].dt, _s())


def conv3x3_mean(partial: torch.Tensor, nsplit, x: Act, w32: torch.Tensor, bias, msum: torch.Tensor, B, H, W, Cout):
    """partial [B, nsplit, C] = channel sums of x (channel_sum); w32 [Cout, 3*3*C] float32 in (kh, kw, c) order;
    msum [B, Cout] f32 = spatial SUM of conv3x3(x) + bias."""
    assert w32.dtype == torch.float32 and tuple(w32.shape) == (Cout, 9 * x.C) and msum.dtype == torch.float32 and msum.numel() >= B * Cout
    hip.call("cfp_conv3x3_mean", partial.data_ptr(), nsplit, x.ptr, x.ld, w32.data_ptr(), hip.ptr(bias), msum.data_ptr(), B, H, W, x.C, Cout, x.dt, _s())


def mbconv_plan(B, H, W, Cin, mid):
    """-> (tiles per image, KP) of cfp_mbconv_expand_dw, or None if the shape does not fit its LDS tile."""
    import ctypes
    t, kp = ctypes.c_int(0), ctypes.c_int(0)
    rc = hip.load().cfp_mbconv_plan(B, H, W, Cin, mid, ctypes.byref(t), ctypes.byref(kp))
    return (t.value, kp.value) if rc == 0 else None


def pack_mbconv_pw(w2d: torch.Tensor, dtype) -> torch.Tensor:
    """Expand weights [mid, Cin] (values already on the 16-bit grid or float32) -> [mid, KP + 8] in `dtype`, zero padded: the LDS image
    layout cfp_mbconv_expand_dw copies linearly (KP = Cin rounded up to 32)."""
    mid, cin = w2d.shape
    kp = (cin + 31) // 32 * 32
    out = torch.zeros(mid, kp + 8, dtype=dtype)
    out[:, :cin] = w2d.to(dtype)
    return out.contiguous()


def mbconv_expand_dw(x: Act, wpw: torch.Tensor, s1, t1, wdw: torch.Tensor, s2, t2, out: Act, partial: Optional[torch.Tensor], B, H, W):
    mid = out.C
    assert wpw.dtype == x.buf.dtype and wpw.shape[0] == mid and wdw.shape == (9, mid) and wdw.dtype == x.buf.dtype
    hip.call("cfp_mbconv_expand_dw", x.ptr, x.ld, wpw.data_ptr(), s1.data_ptr(), t1.data_ptr(), wdw.data_ptr(), s2.data_ptr(), t2.data_ptr(),
             out.ptr, out.ld, hip.ptr(partial), B, H, W, x.C, mid, x.dt, _s())


def lkpm_tail(t: Act, xin: Act, out: Act, w1, b1, w2, b2, ln_g, ln_b, rows, ln_eps=1e-6):
    D = t.C
    x3 = t.buf.dtype == torch.float32 and w1.dtype == torch.float16      # float32 tensors + pre-split f16x3 operands (pack_w_x3)
    k1, k2 = (2 * D, 8 * D) if x3 else (D, 4 * D)
    assert xin.C == D and out.C == D and tuple(w1.shape) == (4 * D, k1) and tuple(w2.shape) == (D, k2) and (x3 or w1.dtype == t.buf.dtype)
    hip.call("cfp_lkpm_tail", t.ptr, t.ld, xin.ptr, xin.ld, out.ptr, out.ld, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
             ln_g.data_ptr(), ln_b.data_ptr(), float(ln_eps), rows, D, hip.F32X3 if x3 else t.dt, _s())
