"""CFPNet forward + backward in training mode (batch-statistics BatchNorm) on the HIP tape.

Mirrors `Deltar.forward` in `model.train()` (`src/models/deltar.py:34-67` and the modules it calls: encoder.py,
decoder.py, fusion.py, transformer.py, convnext.py, attention.py) followed by `SILogLoss` and `loss.backward()`
(`train.py:119-125`).  Everything numeric is a `cfp_*` kernel through `autograd_hip.Tape`; this file is the launch
list.  Float32 storage (the parity mode); there is no CPU path.
"""
from __future__ import annotations

import os

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip, ops, spec, train_ops
from .autograd_hip import P, Tape, V
from .geometry import FusionGeometry, gsa_keys, lsa_padding

ENC_STAGES = [("conv0.2", 1), ("conv1", 2), ("conv2", 2), ("conv3.0", 2), ("conv3.1", 1), ("conv4", 2)]
ENC_EPS, ENC_MOM = 1e-3, 0.01          # timm tf_ models: bn_eps 1e-3, bn_momentum 1 - 0.99
TWINS_HEADS, X2I_HEADS = 8, 4


def same_pad(size: int, k: int, s: int) -> Tuple[int, int]:
    total = max((math.ceil(size / s) - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def _pad_cols(t2d: torch.Tensor, cin: int, cin_pad: int, k2: int) -> torch.Tensor:
    if cin == cin_pad:
        return t2d
    out = torch.zeros(t2d.shape[0], k2, cin_pad, dtype=t2d.dtype, device=t2d.device)
    out[:, :, :cin] = t2d.reshape(t2d.shape[0], k2, cin)
    return out.reshape(t2d.shape[0], k2 * cin_pad)


class TrainNet:
    # tape marks in the order the backward reaches them (`forward_backward(stop=...)`, `finish_backward(stop=...)`): the trainer cuts
    # the captured step there
    BACKWARD_MARKS = ("encoder",)

    """Parameters of the reference's state_dict as tape parameters + the training-mode forward/backward."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], layer_names: Sequence[str], device="cuda:0", n_bins=256, min_val=1e-3,
                 max_val=10.0, stem_act: bool = False, change_embedding: bool = True, share_buffers: bool = False, dtype=torch.float32,
                 no_skip_inside: bool = False, norm: str = "linear", base_resolution=spec.BASE_RESOLUTION):
        """`dtype`: storage of activations and of the matrix-core weight operands (float32 = parity mode; bfloat16 /
        float16 = mixed precision: float32 master parameters, float32 gradients of the parameters, 16-bit activations
        and activation gradients, float32 accumulation everywhere)."""
        self.dev = torch.device(device)
        self.dtype = dtype
        self.base_resolution = tuple(base_resolution)
        self.fusion = spec.fusion_table(self.base_resolution)
        self.layers = list(layer_names)
        self.n_bins, self.min_val, self.max_val = n_bins, min_val, max_val
        self.stem_act, self.change_embedding, self.no_skip_inside = stem_act, change_embedding, no_skip_inside
        if norm != "linear":
            raise NotImplementedError(f"bin-width normalisation '{norm}': only 'linear' (every shipped config) has a training path")
        # parameters live on the device in the reference's layout (possibly as views of a flat optimizer buffer, see
        # trainer.Trainer); the kernel layouts are re-derived from them at every step (`zero_grad` drops the derived copies)
        self.sd = {k: (v.detach().float().to(self.dev) if v.is_floating_point() else v) for k, v in state_dict.items()}
        self.P: Dict[str, P] = {}
        self.buf: Dict[str, torch.Tensor] = {}          # running statistics (updated in place)
        for k, v in self.sd.items():
            if k.endswith(("running_mean", "running_var")):
                self.buf[k] = v if share_buffers else v.clone()       # share_buffers: update the caller's running statistics in place
        self._idx_cache: Dict = {}
        self.res_fused = os.environ.get("CFP_RES_FUSED_TRAIN", "1") != "0"    # skip connections added inside the BatchNorm-apply / LayerNorm pass
        self.se_fused = os.environ.get("CFP_SE_FUSED_TRAIN", "1") != "0"      # squeeze-excite gate + its backward as three kernels (csrc/se_train.hip)
        self.side_stream: Optional[torch.cuda.Stream] = None      # set by the trainer: parameter gradients beside the dY -> dX chain
        self._open_tape = None
        self.flipped: Dict[str, torch.Tensor] = {}        # bound mode: name -> flipped conv weight, refreshed by the trainer every step
        self._bound = None                                # (FlatParams in kernel layouts, 16-bit shadow) once `bind` was called
        self.discovered: Optional[Dict[str, tuple]] = None    # set to {} to collect name -> (float32 kernel layout, to_torch, is16)
        self.record: Optional[Dict[str, V]] = None       # debugging: name -> tape value (tools/train_grad_check.py compares their .g)

    # ------------------------------------------------------------------ parameters
    def _param(self, name: str, make32, to_torch, is16: bool) -> P:
        """Tape parameter `name`.  Unbound (default): the kernel layout is derived from the reference-layout tensor in `self.sd`
        by `make32()` (float32) and cast to the storage dtype when `is16`.  Bound (`bind`): the parameter IS a view of the
        trainer's kernel-layout buffers -- nothing is derived, and its gradient is written in place."""
        p = self.P.get(name)
        if p is None:
            if self._bound is not None:
                flat, shadow = self._bound
                s = flat._by_name[name]
                src = shadow if (is16 and shadow is not None) else flat.param
                p = P(name, src[s.start:s.start + s.numel].view(s.shape), to_torch, gview=flat.view(name, "grad"))
                p.wt = self.flipped.get(name)
            else:
                t32 = make32().contiguous().to(self.dev)
                if self.discovered is not None:
                    self.discovered[name] = (t32, to_torch, is16)
                p = P(name, t32.to(self.dtype) if is16 else t32, to_torch)
            self.P[name] = p
        return p

    def bind(self, flat, shadow: Optional[torch.Tensor]) -> None:
        """Read the parameters from `flat.param` (float32, kernel layouts, see trainer.Trainer) -- the 16-bit operands from
        `shadow`, its per-step cast copy -- and write their gradients into `flat.grad`."""
        self._bound = (flat, shadow)
        self.P = {}

    def _conv_w(self, name: str, cin_pad: Optional[int] = None, cout_pad: Optional[int] = None, f32: bool = False) -> P:
        """[Cout, Cin, kh, kw] / [Cout, Cin(, 1)] -> [Cout_pad, kh*kw*Cin_pad]; padding rows / columns are zero and their
        gradients are dropped (channel counts the 16-byte vectors cannot express: RGB 3, ToF 1, squeeze-excite 34 / 58)."""
        if name in self.P:
            return self.P[name]
        shape = tuple(self.sd[name].shape)
        co, ci = shape[0], shape[1]
        kh, kw = (shape[2], shape[3]) if len(shape) == 4 else ((shape[2], 1) if len(shape) == 3 else (1, 1))
        cp, cop = cin_pad or ci, cout_pad or co

        def make32():
            w = self.sd[name].detach().float().reshape(co, ci, kh, kw)
            t = torch.zeros(cop, kh * kw * cp, device=w.device)
            t[:co] = _pad_cols(w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci), ci, cp, kh * kw)
            return t

        def back(g):
            return g[:co].reshape(co, kh, kw, cp)[..., :ci].permute(0, 3, 1, 2).reshape(shape)
        return self._param(name, make32, back, not f32)     # f32: layers fed by float32 [B, C] vectors

    def _vec(self, name: str, pad_to: Optional[int] = None) -> P:
        if name in self.P:
            return self.P[name]
        shape = tuple(self.sd[name].shape)
        n = self.sd[name].numel()

        def make32():
            v = self.sd[name].detach().float().reshape(-1)
            return torch.cat([v, torch.zeros(pad_to - n, device=v.device)]) if (pad_to and pad_to > n) else v.clone()
        return self._param(name, make32, lambda g: g[:n].reshape(shape), False)

    def _dw3(self, name: str) -> P:
        if name in self.P:
            return self.P[name]
        C = self.sd[name].shape[0]
        return self._param(name, lambda: self.sd[name].detach().float().reshape(C, 9).t(), lambda g: g.t().reshape(C, 1, 3, 3), True)

    def _dwl(self, name: str) -> P:
        if name in self.P:
            return self.P[name]
        return self._param(name, lambda: self.sd[name].detach().float()[:, 0].clone(), lambda g: g.unsqueeze(1), False)

    def _table(self, name: str) -> P:
        if name in self.P:
            return self.P[name]
        return self._param(name, lambda: self.sd[name].detach().float().clone(), lambda g: g, False)

    def grads(self) -> Dict[str, torch.Tensor]:
        """Gradients in the reference's state_dict layout (float32, on the device); parameters the forward never touched
        (the 48 dead tensors) are absent, like `p.grad is None` in the reference."""
        return {n: p.to_torch(p.g) for n, p in self.P.items() if p.g is not None}

    def zero_grad(self):
        """Drop the gradients AND the kernel-layout copies of the parameters (they are rebuilt from `self.sd`, which the
        optimizer has updated in place -- or, when bound, re-viewed from the trainer's buffers)."""
        self.P = {}

    # ------------------------------------------------------------------ building blocks
    def _bn(self, t: Tape, x: V, prefix: str, act: int, eps=1e-5, mom=0.1, residual: Optional[V] = None) -> V:
        return t.bn_act(x, self._vec(prefix + ".weight"), self._vec(prefix + ".bias"), self.buf[prefix + ".running_mean"],
                        self.buf[prefix + ".running_var"], eps, mom, act, residual=residual)

    def _conv_same(self, t: Tape, x: V, wname: str, B, H, W, stride, cin_pad=None, bias: Optional[str] = None, stats: bool = False):
        w = self._conv_w(wname, cin_pad)
        k = self.sd[wname].shape[-1]
        Ho, Wo = -(-H // stride), -(-W // stride)
        pt, _ = same_pad(H, k, stride)
        pl, _ = same_pad(W, k, stride)
        return t.conv(x, w, self._vec(bias) if bias else None, B, H, W, k, stride, pt, pl, Ho, Wo, stats=stats), Ho, Wo

    def _conv3(self, t: Tape, x: V, wname: str, bias: Optional[str], B, H, W, stats: bool = False) -> V:
        return t.conv(x, self._conv_w(wname), self._vec(bias) if bias else None, B, H, W, 3, 1, 1, 1, H, W, stats=stats)

    def _encoder(self, t: Tape, x: V, B, H, W) -> List[Tuple[V, int, int]]:
        p = "img_encoder"
        x, H, W = self._conv_same(t, x, f"{p}.conv0.0.weight", B, H, W, 2, cin_pad=8, stats=True)
        x = self._bn(t, x, f"{p}.conv0.1", hip.ACT_SILU if self.stem_act else hip.ACT_NONE, ENC_EPS, ENC_MOM)
        taps = []
        for stage, stride in ENC_STAGES:
            i = 0
            while any(k.startswith(f"{p}.{stage}.{i}.") for k in self.sd):
                q = f"{p}.{stage}.{i}"
                s = stride if i == 0 else 1
                inp, Hi, Wi = x, H, W
                if f"{q}.conv.weight" in self.sd:
                    x, H, W = self._conv_same(t, x, q + ".conv.weight", B, H, W, s, stats=True)
                    x = self._bn(t, x, q + ".bn1", hip.ACT_SILU, ENC_EPS, ENC_MOM)
                elif f"{q}.conv_exp.weight" in self.sd:
                    x, H, W = self._conv_same(t, x, q + ".conv_exp.weight", B, H, W, s, stats=True)
                    x = self._bn(t, x, q + ".bn1", hip.ACT_SILU, ENC_EPS, ENC_MOM)
                    x = t.conv(x, self._conv_w(q + ".conv_pwl.weight"), None, B, H, W, 1, 1, 0, 0, H, W, stats=True)
                    skip = inp if (self.res_fused and s == 1 and inp.C == x.C) else None       # the skip connection rides on the last BatchNorm's apply pass
                    x = self._bn(t, x, q + ".bn2", hip.ACT_NONE, ENC_EPS, ENC_MOM, residual=skip)
                    if skip is not None:
                        inp = None
                else:
                    x = t.conv(x, self._conv_w(q + ".conv_pw.weight"), None, B, H, W, 1, 1, 0, 0, H, W, stats=True)
                    x = self._bn(t, x, q + ".bn1", hip.ACT_SILU, ENC_EPS, ENC_MOM)
                    Ho, Wo = -(-H // s), -(-W // s)
                    x = t.dw3x3(x, self._dw3(q + ".conv_dw.weight"), B, H, W, s, same_pad(H, 3, s)[0], same_pad(W, 3, s)[0], Ho, Wo)
                    H, W = Ho, Wo
                    x = self._bn(t, x, q + ".bn2", hip.ACT_SILU, ENC_EPS, ENC_MOM)
                    R = self.sd[q + ".se.conv_reduce.weight"].shape[0]
                    Rp = -(-R // 4) * 4                               # zero-padded hidden units: SiLU(0) = 0 feeds zero columns
                    w1, b1 = self._conv_w(q + ".se.conv_reduce.weight", cout_pad=Rp, f32=True), self._vec(q + ".se.conv_reduce.bias", Rp)
                    w2, b2 = self._conv_w(q + ".se.conv_expand.weight", cin_pad=Rp, f32=True), self._vec(q + ".se.conv_expand.bias")
                    if self.se_fused and Rp <= 64 and x.C <= 2048 and x.C % 4 == 0 and B <= 64:
                        x = t.se_block(x, w1, b1, w2, b2, B, H * W)    # 3 + 4 launches instead of ~8 + ~16
                    else:
                        g = t.channel_mean(x, B, H * W)
                        g = t.act(t.linear(g, w1, b1), hip.ACT_SILU)
                        g = t.act(t.linear(g, w2, b2), hip.ACT_SIGMOID)
                        x = t.mul_bcast(x, g, B, H * W)
                    x = t.conv(x, self._conv_w(q + ".conv_pwl.weight"), None, B, H, W, 1, 1, 0, 0, H, W, stats=True)
                    skip = inp if (self.res_fused and s == 1 and inp.C == x.C) else None
                    x = self._bn(t, x, q + ".bn3", hip.ACT_NONE, ENC_EPS, ENC_MOM, residual=skip)
                    if skip is not None:
                        inp = None
                if inp is not None and s == 1 and inp.C == x.C:
                    x = t.add(x, inp)
                i += 1
            if stage != "conv3.0":
                taps.append((x, H, W))
        return taps

    def _hist_encoder(self, t: Tape, hist: torch.Tensor) -> List[V]:
        B, Z, N = hist.shape
        rows = B * Z * N
        x8 = ops.new_act(rows, 8, self.dtype, self.dev)
        ops.scalar_to_rows8(hist.reshape(-1).contiguous(), x8, rows)
        x = V(x8.buf, needs_grad=False)
        outs = []
        for e in (1, 2, 3):
            q = f"hist_encoder.hist_extractor{e}.pointnet_encoder"
            for j in (1, 2, 3):
                x = t.linear(x, self._conv_w(f"{q}.conv{j}.weight", cin_pad=8 if (e == 1 and j == 1) else None), self._vec(f"{q}.conv{j}.bias"), stats=True)
                x = self._bn(t, x, f"{q}.bn{j}", hip.ACT_RELU)
            outs.append(x)
        return outs

    def _loftr(self, t: Tape, p: str, x: V, src: V, N, L, S, heads) -> V:
        D = x.C
        q = t.linear(x, self._conv_w(p + ".q_proj.weight"))
        k = t.linear(src, self._conv_w(p + ".k_proj.weight"))
        v = t.linear(src, self._conv_w(p + ".v_proj.weight"))
        msg = t.attention(q, k, v, N, L, S, heads, D // heads)
        msg = t.linear(msg, self._conv_w(p + ".merge.weight"))
        msg = t.layernorm(msg, self._vec(p + ".norm1.weight"), self._vec(p + ".norm1.bias"), 1e-5)
        h = t.linear(t.concat(x, msg), self._conv_w(p + ".mlp.0.weight"))
        h = t.linear(t.act(h, hip.ACT_RELU), self._conv_w(p + ".mlp.2.weight"))
        if self.res_fused:
            return t.layernorm(h, self._vec(p + ".norm2.weight"), self._vec(p + ".norm2.bias"), 1e-5, residual=x)      # norm2(mlp) + x
        return t.add(t.layernorm(h, self._vec(p + ".norm2.weight"), self._vec(p + ".norm2.bias"), 1e-5), x)

    def _maps(self, key, build):
        if key not in self._idx_cache:
            idx, n_src = build()
            idx = torch.as_tensor(np.ascontiguousarray(idx), dtype=torch.int32).to(self.dev)
            self._idx_cache[key] = (idx, train_ops.inverse_index(idx, n_src))
        return self._idx_cache[key]

    def _lsa(self, t: Tape, p: str, tok: V, B, H, W, ws) -> V:
        pb, pr = lsa_padding(H, W, ws)
        Hp, Wp = H + pb, W + pr
        nh, nw = Hp // ws, Wp // ws

        def build():
            b, a, c, i, j = np.meshgrid(np.arange(B), np.arange(nh), np.arange(nw), np.arange(ws), np.arange(ws), indexing="ij")
            y, x = a * ws + i, c * ws + j
            idx = np.where((y < H) & (x < W), (b * H + y) * W + x, -1)
            return idx.reshape(-1), B * H * W
        idx, inv = self._maps(("lsa", B, H, W, ws), build)
        x = t.gather(tok, idx, inv)                                         # window partition with zero padding
        x = self._loftr(t, p + ".encoder_layer", x, x, B * nh * nw, ws * ws, ws * ws, TWINS_HEADS)
        return t.gather(x, inv, idx)                                        # un-partition, padded rows dropped

    def _gsa(self, t: Tape, p: str, tok: V, B, H, W, ws) -> V:
        Hk, Wk = gsa_keys(H, W, ws)
        x = t.conv(tok, self._conv_w(p + ".sr.weight"), self._vec(p + ".sr.bias"), B, H, W, ws, ws, 0, 0, Hk, Wk)
        x = t.layernorm(x, self._vec(p + ".norm.weight"), self._vec(p + ".norm.bias"), 1e-5)
        return self._loftr(t, p + ".encoder_layer", tok, x, B, H * W, Hk * Wk, TWINS_HEADS)

    def _dapm(self, t: Tape, p: str, tok: V, B, H, W, rect, heads) -> V:
        y0, y1, x0, x1 = rect
        D = tok.C

        def build_in():
            b, y, x = np.meshgrid(np.arange(B), np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
            return ((b * H + y) * W + x).reshape(-1), B * H * W
        idx_in, inv_in = self._maps(("dapm_in", B, H, W, rect), build_in)

        def build_out():                                               # identity with the inside rows zeroed
            b, y, x = np.meshgrid(np.arange(B), np.arange(H), np.arange(W), indexing="ij")
            inside = (y >= y0) & (y < y1) & (x >= x0) & (x < x1)
            return np.where(inside, -1, (b * H + y) * W + x).reshape(-1), B * H * W
        idx_out, inv_out = self._maps(("dapm_out", B, H, W, rect), build_out)
        inside = t.gather(tok, idx_in, inv_in)
        q = t.linear(tok, self._conv_w(p + ".q_proj.weight"))
        k = t.linear(inside, self._conv_w(p + ".k_proj.weight"))
        v = t.linear(inside, self._conv_w(p + ".v_proj.weight"))
        S = (y1 - y0) * (x1 - x0)
        msg = t.attention(q, k, v, B, H * W, S, heads, D // heads)
        msg = t.gather(msg, idx_out, inv_out)                              # only outside tokens receive a message
        f = t.concat(tok, msg)
        f = self._bn(t, self._conv3(t, f, p + ".conv1.weight", None, B, H, W, stats=True), p + ".bn1", hip.ACT_NONE)
        if self.res_fused:
            return self._bn(t, self._conv3(t, f, p + ".conv2.weight", None, B, H, W, stats=True), p + ".bn2", hip.ACT_NONE, residual=tok)     # bn2(conv2) + feat
        return t.add(self._bn(t, self._conv3(t, f, p + ".conv2.weight", None, B, H, W, stats=True), p + ".bn2", hip.ACT_NONE), tok)

    def _lkpm(self, t: Tape, p: str, tok: V, B, H, W) -> V:
        k = self.sd[p + ".dwconv2.weight"].shape[-1]
        y = t.dwlarge(tok, self._dwl(p + ".dwconv2.weight"), self._vec(p + ".dwconv2.bias"), B, H, W, k)
        y = self._bn(t, y, p + ".bn1", hip.ACT_RELU)
        y = t.layernorm(y, self._vec(p + ".norm.weight"), self._vec(p + ".norm.bias"), 1e-6)
        y = t.act(t.linear(y, self._conv_w(p + ".pwconv1.weight"), self._vec(p + ".pwconv1.bias")), hip.ACT_GELU)
        y = t.linear(y, self._conv_w(p + ".pwconv2.weight"), self._vec(p + ".pwconv2.bias"))
        return t.add(tok, y)

    def _fusion(self, t: Tape, name: str, x: V, feat1: V, mask: torch.Tensor, patch_info, B, H, W, Z, N, pos_offset) -> V:
        p = f"decoder.{name}"
        D, (Hm, Wm), _ = self.fusion[name]
        ws = spec.window_size((Hm, Wm))
        geo = FusionGeometry.from_patch_info(patch_info, self.base_resolution[1] / Wm)
        zn, p1, p2 = geo.zone_num, geo.p1, geo.p2
        sy, sx, ey, ex = geo.sy_wo, geo.sx_wo, geo.ey_wo, geo.ex_wo
        tzh, tzw = geo.tzh, geo.tzw
        cy0, cy1, cx0, cx1 = geo.clipped(H, W)
        emb0 = t.add_table(x, self._table(p + ".positional_encodings"), B, H, W, Hm, Wm, pos_offset)
        tok = emb0
        src = t.add_table(feat1, self._table(p + ".positional_encodings2"), B * Z, 1, N, 1, N, (0, 0))
        # zone validity as a per-(zone, channel) multiplier built on the device: no host round trip, so the step can be captured
        valid_gate = V(mask.reshape(-1, 1).to(self.dev, torch.float32).expand(B * Z, D).contiguous(), needs_grad=False)
        for i, lname in enumerate(self.layers):
            q = f"{p}.layers.{i}"
            if lname == "image":
                tok = self._lsa(t, q + ".lga", tok, B, H, W, ws)
                tok = self._gsa(t, q + ".gsa", tok, B, H, W, ws)
            elif lname == "hist2image":
                grid = tok if self.change_embedding else emb0

                def build_crop():                                       # rectangle [sy:ey, sx:ex] of the zero-extended map
                    b, y, xx = np.meshgrid(np.arange(B), np.arange(sy, ey), np.arange(sx, ex), indexing="ij")
                    ok = (y >= 0) & (y < H) & (xx >= 0) & (xx < W)
                    return np.where(ok, (b * H + y) * W + xx, -1).reshape(-1), B * H * W
                idx_c, inv_c = self._maps(("crop", B, H, W, sy, ey, sx, ex), build_crop)
                z = t.gather(grid, idx_c, inv_c)
                if geo.interpolate:
                    z = t.resize(z, B, tzh, tzw, zn * p1, zn * p2)

                def build_zone():                                       # (b, zy, zx, i, j) <- (b, zy*p1 + i, zx*p2 + j)
                    b, zy, zx, ii, jj = np.meshgrid(np.arange(B), np.arange(zn), np.arange(zn), np.arange(p1), np.arange(p2), indexing="ij")
                    return ((b * zn * p1 + zy * p1 + ii) * (zn * p2) + zx * p2 + jj).reshape(-1), B * zn * p1 * zn * p2
                idx_z, inv_z = self._maps(("zone", B, zn, p1, p2), build_zone)
                z = t.gather(z, idx_z, inv_z)
                z = self._loftr(t, q, z, src, B * zn * zn, p1 * p2, N, X2I_HEADS)
                z = t.mul_bcast(z, valid_gate, B * zn * zn, p1 * p2)           # zero the zones without a ToF signal
                z = t.gather(z, inv_z, idx_z)                                   # back to the map layout
                if geo.interpolate:
                    z = t.resize(z, B, zn * p1, zn * p2, tzh, tzw)

                def build_paste():                                      # image pixel <- rectangle pixel (part inside the image)
                    b, y, xx = np.meshgrid(np.arange(B), np.arange(H), np.arange(W), indexing="ij")
                    ins = (y >= cy0) & (y < cy1) & (xx >= cx0) & (xx < cx1)
                    return np.where(ins, (b * tzh + (y - sy)) * tzw + (xx - sx), -1).reshape(-1), B * tzh * tzw
                idx_p, inv_p = self._maps(("paste", B, H, W, sy, sx, tzh, tzw, cy0, cy1, cx0, cx1), build_paste)
                if self.no_skip_inside:                                 # fusion.py:156: the rectangle is REPLACED, not added to

                    def build_keep():
                        b, y, xx = np.meshgrid(np.arange(B), np.arange(H), np.arange(W), indexing="ij")
                        ins = (y >= cy0) & (y < cy1) & (xx >= cx0) & (xx < cx1)
                        return np.where(ins, -1, (b * H + y) * W + xx).reshape(-1), B * H * W
                    idx_k, inv_k = self._maps(("keep_out", B, H, W, cy0, cy1, cx0, cx1), build_keep)
                    tok = t.gather(tok, idx_k, inv_k)
                tok = t.add(tok, t.gather(z, idx_p, inv_p))
            elif lname == "combine1":
                tok = self._dapm(t, q + ".transformer_path", tok, B, H, W, (cy0, cy1, cx0, cx1), 4)
                tok = self._lkpm(t, q + ".large_kernel_path", tok, B, H, W)
            else:
                raise NotImplementedError(lname)
            if self.record is not None:
                self.record[q] = tok
        return tok

    def _up(self, t: Tape, p: str, x: V, Hs, Ws, skip: V, B, H, W) -> V:
        x = t.resize(x, B, Hs, Ws, H, W)
        x = t.concat(x, skip)
        for c, b in ((0, 1), (3, 4)):
            x = self._conv3(t, x, f"{p}._net.{c}.weight", f"{p}._net.{c}.bias", B, H, W, stats=True)
            x = self._bn(t, x, f"{p}._net.{b}", hip.ACT_LRELU)
        return x

    # ------------------------------------------------------------------ the step
    def forward(self, t: Tape, input_data: dict, pos_offsets: Optional[dict] = None):
        """Training-mode forward on tape `t`: -> (pred as a tape value [B*h*w, 1], bin edges [B, n_bins+1], (B, h, w))."""
        dev = self.dev
        rgb = input_data["rgb"].to(dev, torch.float32).contiguous()
        add = input_data["additional"]
        B, _, H, W = rgb.shape
        pos_offsets = pos_offsets or {}
        x8 = ops.new_act(B * H * W, 8, self.dtype, dev)
        ops.rgb_to_nhwc8(rgb, x8, B, H, W)
        taps = self._encoder(t, V(x8.buf, needs_grad=False), B, H, W)
        t.mark("encoder")             # backward of everything below runs before the encoder's (trainer: gradient buckets overlap it)
        (b0, h0, w0), (b1, h1, w1), (b2, h2, w2), (b3, h3, w3), (b4, h4, w4) = taps
        hist = add["hist_data"].to(dev, torch.float32).contiguous()
        Z, N = hist.shape[1], hist.shape[2]
        f1, f2, f3 = self._hist_encoder(t, hist)
        mask = add["mask"]
        pinfo = add["patch_info"]

        def pw(x, name, Hh, Ww):
            return t.conv(x, self._conv_w(name + ".weight"), self._vec(name + ".bias"), B, Hh, Ww, 1, 1, 0, 0, Hh, Ww)

        def fuse(name, x, feat, Hh, Ww):
            return self._fusion(t, name, x, feat, mask, pinfo, B, Hh, Ww, Z, N, pos_offsets.get(name, (0, 0)))

        x = pw(b4, "decoder.conv4", h4, w4)
        x = self._up(t, "decoder.up1", x, h4, w4, b3, B, h3, w3)
        x = pw(x, "decoder.conv3", h3, w3)
        x = t.concat(x, fuse("cross_atten3", x, f3, h3, w3))
        x = self._up(t, "decoder.up2", x, h3, w3, b2, B, h2, w2)
        x = pw(x, "decoder.conv2", h2, w2)
        x = t.concat(x, fuse("cross_atten2", x, f2, h2, w2))
        x = self._up(t, "decoder.up3", x, h2, w2, b1, B, h1, w1)
        x = pw(x, "decoder.conv1", h1, w1)
        x = t.concat(x, fuse("cross_atten1", x, f1, h1, w1))
        x = self._up(t, "decoder.up4", x, h1, w1, b0, B, h0, w0)
        unet = self._conv3(t, x, "decoder.conv0.weight", "decoder.conv0.bias", B, h0, w0)
        # adaptive-bin head (decoder.py:22-37, deltar.py:50-61)
        ram = self._conv3(t, unet, "depth_head.conv3x3.weight", "depth_head.conv3x3.bias", B, h0, w0)
        y = t.conv(unet, self._conv_w("depth_head.conv1x1.weight"), None, B, h0, w0, 1, 1, 0, 0, h0, w0)
        y = t.channel_mean(y, B, h0 * w0)
        y = t.act(t.linear(y, self._conv_w("depth_head.regressor.0.weight", f32=True), self._vec("depth_head.regressor.0.bias")), hip.ACT_LRELU)
        y = t.act(t.linear(y, self._conv_w("depth_head.regressor.2.weight", f32=True), self._vec("depth_head.regressor.2.bias")), hip.ACT_LRELU)
        y = t.linear(y, self._conv_w("depth_head.regressor.4.weight", f32=True), self._vec("depth_head.regressor.4.bias"))
        y = t.add_const(t.act(y, hip.ACT_RELU), torch.full((self.n_bins,), 0.1, dtype=torch.float32, device=dev))     # norm == 'linear'
        wn = t.row_normalize(y)
        edges, centers = t.bin_centers(wn, self.min_val, self.max_val)
        logits = pw(ram, "conv_out.0", h0, w0)
        pred = t.softmax_expect(logits, centers, B, h0 * w0)
        return pred, edges, (B, h0, w0)

    def forward_backward(self, input_data: dict, target: torch.Tensor, loss_mask: Optional[torch.Tensor] = None, pos_offsets: Optional[dict] = None,
                         stop_before_encoder: bool = False, loss_sync=None, stop: Optional[str] = None, defer_param_grads: bool = False):
        """One forward in training mode + SILog + backward.  Returns (loss as a device scalar, pred [B,1,H/2,W/2], edges);
        gradients are in `self.grads()`, running statistics in `self.buf`.  `stop_before_encoder`: the backward stops where the
        RGB encoder's begins (every non-encoder parameter gradient is final) and `finish_backward()` runs the rest.
        `loss_sync` = (torch.distributed module, world size): the loss is the global-batch SILog over the ranks (SILogLoss.sync_moments).
        `stop`: any tape mark (BACKWARD_MARKS) instead of "encoder"; `finish_backward(stop=...)` continues to the next one.
        `defer_param_grads`: weight / bias gradient kernels are queued on the tape; `run_deferred_param_grads()` issues them."""
        dev = self.dev
        t = Tape(dev, self.dtype, side=self.side_stream)
        t.defer = defer_param_grads
        pred, edges, (B, h0, w0) = self.forward(t, input_data, pos_offsets)
        # SILog (loss.py:9-19) on the half-resolution prediction against the full-resolution target
        crit = train_ops.SILogLoss()
        pred4 = pred.t.reshape(B, 1, h0, w0)
        loss = crit.forward(pred4, target.to(dev, torch.float32), loss_mask.to(dev) if loss_mask is not None else None, interpolate=True)
        gl = 1.0
        if loss_sync is not None:
            loss = crit.sync_moments(loss_sync[0])
            gl = float(loss_sync[1])                                            # the gradient average over the ranks divides by it again
        pred.g = crit.backward(gl).reshape(-1, 1).contiguous()
        stop = stop or ("encoder" if stop_before_encoder else None)
        t.backward(stop=stop)
        self._open_tape = t if (stop is not None or defer_param_grads) else None
        return loss, pred4, edges

    def finish_backward(self, stop: Optional[str] = None) -> None:
        """The next part of a backward that `forward_backward(stop=...)` left open: down to mark `stop`, or to the end."""
        t = self._open_tape
        t.backward(stop=stop)
        if stop is None and not t.defer:
            self._open_tape = None

    def run_deferred_param_grads(self, last: bool = False) -> list:
        """Issue the parameter-gradient kernels queued so far (`defer_param_grads`); `last`: the tape is closed afterwards."""
        t = self._open_tape
        fns = t.run_deferred()
        if last:
            self._open_tape = None
        return fns
