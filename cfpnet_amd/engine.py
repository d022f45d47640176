"""Inference engine: `Deltar.forward` as a sequence of HIP kernel launches.

The reference's forward (`/root/reference/src/models/deltar.py:34-67`) is ~1000 eager PyTorch
ops.  Here the same computation is a static launch list over NHWC buffers:

  * parameters are packed once (BatchNorm folded into per-channel scale/shift, conv weights
    re-laid [Cout][kh][kw][Cin], q/k/v projections concatenated, storage dtype bf16, f16 or f32);
  * every `rearrange 'b (h w) c <-> b c h w'` of the reference disappears (tokens ARE NHWC);
  * every `torch.cat` disappears: producers write into channel slices of the consumer's buffer;
  * the zone crop / per-zone regrouping / boolean-mask scatter of `fusion.py:103-157` and the
    window partition of `transformer.py:104-115` are kernel addressing (see geometry.py);
  * the whole list can be captured into one HIP graph (`capture()` / `replay()`), because no
    step needs a device->host sync (the reference syncs 6x per forward, transformer.py:218).

torch is used for device memory and the current stream only.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import hip, ops, spec
from .geometry import FusionGeometry, gsa_keys
from .ops import Act

_BN_EPS = 1e-5


def _same_pad(size: int, k: int, s: int) -> Tuple[int, int]:
    total = max((math.ceil(size / s) - 1) * s + k - size, 0)
    return total // 2, total - total // 2


_STREAMS: Dict = {}
_CONCURRENT: Dict[str, list] = {}


def concurrent_streams(device, want: int = 4, pool: int = 16) -> list:
    """Up to `want` HIP streams that the GPU really runs side by side.

    The runtime spreads streams over a handful of hardware queues (4 by default); two streams that share a queue run
    their work back to back however independent it is, and which streams share one is not visible through the API.
    So the pool is PROBED: a spin kernel (`torch.cuda._sleep`) is timed alone and on pairs of streams, and a stream
    joins the set only if it overlaps with every member.  Measured on MI355X with the batch-8 forward graph: 1 / 2 / 3 / 4
    batches in flight on such streams = 4.54 / 3.17 / 2.82 / 2.66 ms per step; two streams on one queue = 4.54."""
    import time
    key = str(torch.device(device))
    got = _CONCURRENT.get(key)
    if got is not None and len(got) >= want:
        return got[:want]
    cand = [torch.cuda.Stream(device=device) for _ in range(pool)]
    cycles = 1_000_000
    if not hasattr(torch.cuda, "_sleep"):          # no spin kernel to probe with: take the streams as they come
        _CONCURRENT[key] = cand[:max(want, 4)]
        return _CONCURRENT[key][:want]

    def spin(ids):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in ids:
            with torch.cuda.stream(cand[i]):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    spin([0])
    base = min(spin([0]) for _ in range(3))
    chosen = [0]
    for j in range(1, pool):
        if len(chosen) >= want:
            break
        if all(min(spin([k, j]) for _ in range(2)) < 1.5 * base for k in chosen):
            chosen.append(j)
    _CONCURRENT[key] = [cand[i] for i in chosen]
    return _CONCURRENT[key][:want]


def _shared_stream(device, role):
    """One HIP stream per (device, role) for the whole process; lane streams come from the probed concurrent set."""
    key = (str(torch.device(device)), role)
    st = _STREAMS.get(key)
    if st is None:
        if isinstance(role, tuple) and role[0] == "lane":
            conc = concurrent_streams(device)
            st = conc[role[1] % len(conc)]
        else:
            st = torch.cuda.Stream(device=device)
        _STREAMS[key] = st
    return st


class Engine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], *, layer_names: Sequence[str], n_bins: int = 256,
                 min_val: float = 1e-3, max_val: float = 10.0, norm: str = "linear", change_embedding: bool = True,
                 no_skip_inside: bool = False, stem_act: bool = False, dtype=None, device="cuda:0",
                 zone_sample_num: int = 16, base_resolution=spec.BASE_RESOLUTION, x3: Optional[bool] = None):
        """`x3` (float32 storage only): every convolution / linear layer runs its matrix math split-precision on the 16-bit matrix cores
        (A_hi W_hi + A_hi W_lo + A_lo W_hi in IEEE half, float32 accumulate; csrc/conv_igemm_x3.hip) instead of the float32 MFMA:
        ~21 significant bits per product at 3/16 instead of 1/16 of the 16-bit matrix rate.  Activations, every other kernel and the
        results' type are those of the float32 mode.  With neither `dtype` nor `x3` given the engine is built in that mode (float32
        storage, x3): the one inside the reference tolerance on every weight family.  torch.float16 / torch.bfloat16 are the opt-in
        16-bit speed modes, torch.float32 without `x3` the bit-level parity mode."""
        if dtype is None:
            dtype, x3 = torch.float32, (True if x3 is None else x3)
        x3 = bool(x3)
        hip.load()   # fail loudly if the HIP extension is missing
        self.base_resolution = tuple(base_resolution)
        self.fusion = spec.fusion_table(self.base_resolution)    # decoder.py:82-94 generalised to other table sizes (configs[4])
        if not torch.cuda.is_available():
            raise RuntimeError("cfpnet_amd.Engine needs a GPU: the product path has no CPU fallback")
        assert dtype in (torch.bfloat16, torch.float16, torch.float32)
        self.layer_names = list(layer_names)
        self.n_bins, self.min_val, self.max_val = n_bins, float(min_val), float(max_val)
        self.norm = {"linear": 0, "softmax": 1, "sigmoid": 2}[norm]
        self.change_embedding, self.no_skip_inside, self.stem_act = change_embedding, no_skip_inside, stem_act
        self.dtype, self.device = dtype, torch.device(device)
        assert not x3 or dtype == torch.float32, "x3 is a matrix-math mode of float32 storage"
        self.x3 = bool(x3)
        self.cdt = hip.F32X3 if self.x3 else ops.DT[dtype]          # planning dtype (cfp_conv2d_ws_bytes)
        # ticketed split-K (CFP_CONV_WS_TICKETS: the last workgroup at a tile finishes it, no reduce launch) is bit-identical and measured EQUAL at
        # batch 1 (3.20 vs 3.18 ms: the memory round trip + ticket it adds to the GEMM is as long as the launch it removes; with the
        # end-of-round single-image plans 2.91 vs 2.86): off
        self.tickets = self.x3 and os.environ.get("CFP_SPLITK_TICKETS", "0") == "1"
        self.zone_sample_num = zone_sample_num
        # depth head: conv3x3 -> conv_out -> softmax -> expectation as ONE kernel (csrc/head_fused.hip) in the 16-bit modes;
        # head_hilo = (conv_out weights as hi + lo planes, ram fed to conv_out as hi + lo): the logits then carry neither the
        # rounding of conv_out's weights nor that of ram -- measured +34 us / +200 us per batch of 8 for < 2 % of the error
        # budget (profiles/r2_precision_budget.md), so both are OFF by default.  CFP_HEAD_FUSED=0 runs the separate kernels.
        # two-term (hi + lo) 16-bit weights for the pointwise layers that run through the gen-2 GEMM with shared weights
        # (ops.pack_w2 / CFP_CONV_W2).  Measured at batch 8 (gpurun_out/ mode_bench, precision_report): fp16 rel-L1 0.90e-3 -> 0.85e-3,
        # bf16 6.7e-3 -> 5.9e-3 for +6.5 % step time -- the layers it cannot reach (per-image project weights, the fused LoFTR tail)
        # keep most of the pointwise rounding error.  OFF by default (CFP_WEIGHTS2=1 turns it on).
        self.weights2 = os.environ.get("CFP_WEIGHTS2", "0") == "1"
        # fp16 only: pointwise (1x1 / Linear) weights are rounded with error diffusion along the INPUT-CHANNEL axis (ops.round_taps on
        # [Cout, 1, Cin]): every row's rounding errors sum to < 1 ulp, so a layer's weight rounding no longer shifts its outputs'
        # means -- the component that adds up coherently from layer to layer.  Measured on the benched batch: rel-L1 1.08e-3 ->
        # 0.80e-3 (every image <= 0.86e-3), free.  In bf16 the same rounding measured worse (weights-only ablation 1.28x: the doubled
        # per-weight variance of direction-only rounding outweighs the cancelled mean shift), so bf16 keeps round-to-nearest there.
        self.diffuse_cin = os.environ.get("CFP_DIFFUSE_CIN", "1") == "1"
        # stride-1 inverted-residual blocks: expand GEMM + depthwise 3x3 (+ SE sums) as one kernel (csrc/mbconv.hip), 16-bit modes.
        # Built, parity-tested and MEASURED SLOWER than the two kernels it replaces (40 vs 40 us alone at 30x40x816, 32 vs 23 us with
        # four copies side by side; whole step 2.96 vs 2.57 ms with its first version): OFF by default, CFP_MBCONV_FUSED=1 enables it.
        self.mbconv_fused = os.environ.get("CFP_MBCONV_FUSED", "0") == "1"
        # squeeze-excite through cfp_dwconv3x3_se_nhwc + cfp_se_gate_fold2 (16-bit modes; CFP_SE2=0: the round-2 pair)
        # The float32 depthwise kernel leaves the same partial dot products, so the default f16x3 mode can take this path too.  Measured:
        # one graph at a time it is faster at every batch (3.69 vs 3.87 ms at batch 1, 4.03 vs 4.26 at 2, 4.91 vs 5.05 at 4, 7.08 vs 7.17 at 8:
        # one launch fewer per block on the critical path), with four batches in flight it is neutral to slightly slower (4.63 vs 4.59 ms at
        # batch 8) -- with the round-1 LDS-strip kernel.  Round 5: dw3x3_rows_kernel (csrc/dw3x3_rows.hip) takes the dot products in its
        # tail at no visible cost, and with four batches in flight the path is now ahead too (4.47 vs 4.51 ms at batch 8): always on
        # (`se2_x3`; CFP_SE2_X3=0 never, =auto outside the in-flight plan only).
        self.se2 = dtype in (torch.bfloat16, torch.float16) and os.environ.get("CFP_SE2", "1") == "1"
        self.se2_x3 = os.environ.get("CFP_SE2_X3", "1") if self.x3 else "0"
        # DIAGNOSTIC ONLY (tools/precision_family.py --acts): "name:dtype,..." rounds the named encoder tensors of a float32 engine to a
        # 16-bit format in place right after they are produced, to attribute the 16-bit error to single tensors.  Never set in product use.
        self._dbg_round = {}
        for item in filter(None, os.environ.get("CFP_DEBUG_ROUND", "").split(",")):
            n, d = item.split(":")
            self._dbg_round[n] = {"f16": torch.float16, "bf16": torch.bfloat16}[d]
        # decoder stages whose bilinear upsample + skip concat run inside the first conv's loader (cfp_upsample_cat_conv3x3); digits 1-4
        self.up_fused = os.environ.get("CFP_UP_FUSED", "4")
        # f16x3 mode: decoder stages whose first conv takes the two-source chunk kernel (cfp_upsample_cat_conv3x3 with CFP_F32X3, round 5).  Built,
        # parity-tested and MEASURED SLOWER than resize + implicit GEMM at every stage (tools/up_bench_x3.py, batch 8, four copies in flight: up1
        # 107 vs 79 us, up2 137 vs 129, up3 219 vs 164, up4 358 vs 225; whole step 4.71 vs 4.68 ms): OFF by default, CFP_UP_FUSED_X3=1234 enables it.
        self.up_fused_x3 = os.environ.get("CFP_UP_FUSED_X3", "")
        self.cat_pad = os.environ.get("CFP_CAT_PAD", "1") == "1" and not self.up_fused_x3      # f16x3: zero-padded concatenation buffers (see _pack: decoder.up*.a)
        self.lkpm_fused = os.environ.get("CFP_LKPM_FUSED", "1") != "0"      # LKPM's LayerNorm + MLP + residual as one kernel (cfp_lkpm_tail)
        self.tail_q = os.environ.get("CFP_TAIL_Q", "1") == "1"          # q projection inside the fused LoFTR tail
        self.lkpm_min_rows = int(os.environ.get("CFP_LKPM_MIN_ROWS", "30000"))      # token rows from which the LKPM tail runs fused
        self.sr_ln_fused = os.environ.get("CFP_SR_LN_FUSED", "1") == "1"      # f16x3: the global attention's LayerNorm inside its patch conv (0: a launch of its own)
        self.head_fused = os.environ.get("CFP_HEAD_FUSED", "1") != "0"
        hl = os.environ.get("CFP_HEAD_HILO", "00")
        self.head_hilo = (hl[0] == "1", hl[1] == "1")
        self.half = dtype in (torch.bfloat16, torch.float16)     # 16-bit storage: the MFMA fast paths
        self.ve = 8 if self.half else 4
        self.P: Dict[str, torch.Tensor] = {}
        self._plans: Dict[tuple, dict] = {}
        self._graph = None
        self._splitk_ws: Optional[torch.Tensor] = None
        self._capturing = False
        # independent branches (ToF histogram encoder beside the RGB encoder, bin-width regressor beside
        # the depth head's 3x3 conv) run on a second HIP stream; fork/join are events, also inside a graph
        self._side = _shared_stream(self.device, "side")
        # batch lanes: the batch can be cut into `lanes` sub-batches whose forwards run concurrently on their own
        # streams inside one graph (see forward_lanes); per-lane scratch, streams and split-K workspaces
        self._lane = 0
        self.use_side_stream = os.environ.get("CFP_NO_SIDE_STREAM", "0") != "1"
        self._lane_streams: Dict[int, Tuple[torch.cuda.Stream, torch.cuda.Stream]] = {}
        self._lanes_active = 1
        self._slots = None
        self._slot_next = 0
        self._lane_ws: Dict[int, torch.Tensor] = {}
        self.load_state_dict(state_dict)

    # ------------------------------------------------------------------------------ packing
    def _dev(self, t: torch.Tensor, dtype=None) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=dtype or torch.float32).contiguous()

    def _pack_conv(self, w: torch.Tensor, w2: bool = False) -> torch.Tensor:
        """[Co,Ci,kh,kw] or [Co,Ci] or [Co,Ci,1] -> [Co, kh*kw*Ci_padded] in the storage dtype.  `w2` (pointwise layers, 16-bit
        modes, self.weights2): two-term rows [hi | lo] instead (ops.pack_w2)."""
        w = w.detach().float()
        if w.dim() == 2:
            w = w[:, :, None, None]
        elif w.dim() == 3:
            w = w[:, :, :, None]
        co, ci, kh, kw = w.shape
        if kh * kw > 1:
            w = ops.round_taps(w, self.dtype)       # 16-bit storage: rounding error diffused over the taps (ops.round_taps)
        cip = (ci + 7) // 8 * 8
        if cip != ci:
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cip - ci))
        flat = w.permute(0, 2, 3, 1).reshape(co, kh * kw * cip)
        if w2 and self.weights2 and self.half and kh * kw == 1:
            return ops.pack_w2(flat, self.dtype).to(self.device)
        if kh * kw == 1 and self.diffuse_cin and self.dtype == torch.float16:
            flat = ops.round_taps(flat.reshape(co, 1, kh * kw * cip), self.dtype).reshape(co, -1)     # error diffusion along the input channels
        if self.x3:
            return ops.pack_w_x3(self._dev(flat))      # pre-split hi | lo halves in the f16x3 kernels' lane order
        return self._dev(flat, self.dtype)

    def _fold_bn(self, sd, bn: Optional[str], bias: Optional[torch.Tensor], co: int, eps: float):
        """y = conv*scale + shift  ==  BN(conv + bias)"""
        if bn is None:
            scale = torch.ones(co)
            shift = bias.detach().float().clone() if bias is not None else torch.zeros(co)
        else:
            g, b = sd[bn + ".weight"].float(), sd[bn + ".bias"].float()
            m, v = sd[bn + ".running_mean"].float(), sd[bn + ".running_var"].float()
            scale = g / torch.sqrt(v + eps)
            shift = b - m * scale
            if bias is not None:
                shift = shift + bias.detach().float() * scale
        return self._dev(scale), self._dev(shift)

    def _conv(self, sd, name: str, wkey: str, bkey: Optional[str] = None, bn: Optional[str] = None, eps: float = _BN_EPS, w2: bool = True):
        w = sd[wkey]
        self.P[name + ".w"] = self._pack_conv(w, w2)
        s, t = self._fold_bn(sd, bn, sd[bkey] if bkey else None, w.shape[0], eps)
        self.P[name + ".s"], self.P[name + ".t"] = s, t

    def _loftr_pack(self, sd, p: str, self_attn: bool):
        D = sd[p + ".q_proj.weight"].shape[0]
        if self_attn:
            self.P[p + ".qkv"] = self._pack_conv(torch.cat([sd[p + ".q_proj.weight"], sd[p + ".k_proj.weight"], sd[p + ".v_proj.weight"]], 0), True)
        else:
            self.P[p + ".q"] = self._pack_conv(sd[p + ".q_proj.weight"], True)
        self.P[p + ".kv"] = self._pack_conv(torch.cat([sd[p + ".k_proj.weight"], sd[p + ".v_proj.weight"]], 0), True)
        self.P[p + ".q1"] = self._pack_conv(sd[p + ".q_proj.weight"])      # single-term copy for the fused tail's own q projection
        self.P[p + ".merge"] = self._pack_conv(sd[p + ".merge.weight"])
        self.P[p + ".mlp0"] = self._pack_conv(sd[p + ".mlp.0.weight"])
        self.P[p + ".mlp2"] = self._pack_conv(sd[p + ".mlp.2.weight"])
        for n in ("norm1", "norm2"):
            self.P[f"{p}.{n}.g"], self.P[f"{p}.{n}.b"] = self._dev(sd[f"{p}.{n}.weight"]), self._dev(sd[f"{p}.{n}.bias"])

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Pack a reference-layout state dict for the kernels."""
        self.P.clear()
        self._graph = None
        e = "img_encoder"
        EPS = spec.ENC_BN_EPS
        self._conv(sd, "stem", f"{e}.conv0.0.weight", bn=f"{e}.conv0.1", eps=EPS)
        # 16-bit modes: the stem reads the image as hi + lo pairs (cfp_rgb_to_nhwc8_hilo), i.e. exactly; its weight rows repeat the three
        # real input channels in the channel slots 3-5 of every tap (slots the K padding to 8 per tap left empty)
        self.rgb_hilo = self.half and os.environ.get("CFP_RGB_HILO", "1") == "1"
        if self.rgb_hilo:
            w = self.P["stem.w"].reshape(-1, 9, 8).clone()
            w[:, :, 3:6] = w[:, :, 0:3]
            self.P["stem.w"] = w.reshape(w.shape[0], 72).contiguous()
        for b in spec.ENC_BLOCKS:
            q = f"{e}.{b.prefix}"
            if b.kind == "cn":
                self._conv(sd, q + ".conv", q + ".conv.weight", bn=q + ".bn1", eps=EPS)
            elif b.kind == "er":
                self._conv(sd, q + ".exp", q + ".conv_exp.weight", bn=q + ".bn1", eps=EPS)
                self._conv(sd, q + ".pwl", q + ".conv_pwl.weight", bn=q + ".bn2", eps=EPS)
            else:
                self._conv(sd, q + ".pw", q + ".conv_pw.weight", bn=q + ".bn1", eps=EPS)
                if self.half and self.mbconv_fused and b.stride == 1 and not self.weights2:
                    # the same rounded expand weights in the LDS image layout of the fused expand -> depthwise kernel
                    self.P[q + ".pw.wimg"] = ops.pack_mbconv_pw(self.P[q + ".pw.w"].float().cpu(), self.dtype).to(self.device)
                wd = ops.round_taps(sd[q + ".conv_dw.weight"].float(), self.dtype)
                self.P[q + ".dw.w"] = self._dev(wd.reshape(wd.shape[0], 9).t(), self.dtype)
                self.P[q + ".dw.s"], self.P[q + ".dw.t"] = self._fold_bn(sd, q + ".bn2", None, wd.shape[0], EPS)
                self.P[q + ".se.wr"] = self._dev(sd[q + ".se.conv_reduce.weight"].reshape(b.se_rd, b.mid))
                self.P[q + ".se.br"] = self._dev(sd[q + ".se.conv_reduce.bias"])
                self.P[q + ".se.we_t"] = self._dev(sd[q + ".se.conv_expand.weight"].reshape(b.mid, b.se_rd).t())
                self.P[q + ".se.be"] = self._dev(sd[q + ".se.conv_expand.bias"])
                self._conv(sd, q + ".pwl", q + ".conv_pwl.weight", bn=q + ".bn3", eps=EPS, w2=False)   # folded per image by se_gate_fold
                # float32 master of the project weights: cfp_se_gate_fold2 rounds (weight x gate) to the storage type ONCE
                self.P[q + ".pwl.w32"] = self._dev(sd[q + ".conv_pwl.weight"].reshape(b.cout, b.mid))
        # ToF histogram encoder: one float32 parameter blob for the fused kernel (csrc/hist_encoder.hip): per layer W | scale | shift
        parts, layout, off = [], [], 0
        for ex in (1, 2, 3):
            q = f"hist_encoder.hist_extractor{ex}.pointnet_encoder"
            for j in (1, 2, 3):
                w = sd[f"{q}.conv{j}.weight"].detach().float()
                w = w.reshape(w.shape[0], -1)
                co, ci = w.shape
                sc, sh = self._fold_bn(sd, f"{q}.bn{j}", sd[f"{q}.conv{j}.bias"], co, _BN_EPS)
                row = []
                for t in (w.reshape(-1), sc.cpu(), sh.cpu()):
                    row.append(off)
                    parts.append(t.reshape(-1).cpu())
                    off += (t.numel() + 3) // 4 * 4
                    if t.numel() % 4:
                        parts.append(torch.zeros(4 - t.numel() % 4))
                layout.append((row[0], row[1], row[2], ci, co))
        self.P["hist.blob"] = self._dev(torch.cat(parts))
        self._hist_layout = layout
        d = "decoder"
        self._conv(sd, d + ".conv4", d + ".conv4.weight", d + ".conv4.bias")
        for i in (1, 2, 3, 4):
            q = f"{d}.up{i}._net"
            if self.x3 and self.cat_pad:
                # f16x3 mode (round 5): the concatenation buffer [upsampled | skip] is padded with zero channels to a multiple of 32 (80 / 168 / 312 /
                # 392 -> 96 / 192 / 320 / 416) and the weights with zero columns, so that the first conv of the stage is a Cin % 32 == 0 problem for
                # the chunk-pipelined 3x3 kernel (two workgroups per CU) instead of an im2col implicit GEMM that fetches its input once per tap
                wfull = sd[q + ".0.weight"]
                cpad = (wfull.shape[1] + 31) // 32 * 32
                self.P[f"{d}.up{i}.a.wpad"] = self._pack_conv(torch.nn.functional.pad(wfull.detach().float(), (0, 0, 0, 0, 0, cpad - wfull.shape[1])))
            self._conv(sd, f"{d}.up{i}.a", q + ".0.weight", q + ".0.bias", bn=q + ".1")
            if self.x3 and str(i) in self.up_fused_x3:
                # the same weights packed over the PADDED concatenation axis [upsampled decoder channels | skip channels -> next multiple of 32]:
                # the operand of cfp_upsample_cat_conv3x3 in the f16x3 mode (two-source chunk kernel: no resize launch, no concatenation)
                wfull = sd[q + ".0.weight"].detach().float()                  # [Cout, Cup + Cskip, 3, 3]
                cup = wfull.shape[1] - spec.DEC_ENC_CH[i]
                self.P[f"{d}.up{i}.a.wcat"] = ops.pack_w_x3_cat(self._dev(wfull.permute(0, 2, 3, 1).contiguous()), cup)
            self._conv(sd, f"{d}.up{i}.b", q + ".3.weight", q + ".3.bias", bn=q + ".4")
        for n in ("conv3", "conv2", "conv1", "conv0"):
            self._conv(sd, f"{d}.{n}", f"{d}.{n}.weight", f"{d}.{n}.bias")
        for name in self.fusion:
            q = f"{d}.{name}"
            self.P[q + ".pe"] = self._dev(sd[q + ".positional_encodings"])
            self.P[q + ".pe2"] = self._dev(sd[q + ".positional_encodings2"])
            for i, ln in enumerate(self.layer_names):
                l = f"{q}.layers.{i}"
                if ln == "hist2image":
                    self._loftr_pack(sd, l, False)
                elif ln == "image":
                    self._loftr_pack(sd, l + ".lga.encoder_layer", True)
                    self._loftr_pack(sd, l + ".gsa.encoder_layer", False)
                    self._conv(sd, l + ".gsa.sr", l + ".gsa.sr.weight", l + ".gsa.sr.bias")
                    self.P[l + ".gsa.norm.g"], self.P[l + ".gsa.norm.b"] = self._dev(sd[l + ".gsa.norm.weight"]), self._dev(sd[l + ".gsa.norm.bias"])
                elif ln == "combine1":
                    t = l + ".transformer_path"
                    self.P[t + ".qkv"] = self._pack_conv(torch.cat([sd[t + ".q_proj.weight"], sd[t + ".k_proj.weight"], sd[t + ".v_proj.weight"]], 0), True)
                    self._conv(sd, t + ".conv1", t + ".conv1.weight", bn=t + ".bn1")
                    self._conv(sd, t + ".conv2", t + ".conv2.weight", bn=t + ".bn2")
                    k = l + ".large_kernel_path"
                    wd = sd[k + ".dwconv2.weight"].float()
                    kk = wd.shape[-1]
                    self.P[k + ".dw.w"] = self._dev(wd[:, 0].transpose(1, 2).reshape(wd.shape[0], kk * kk))   # f32 [C][kx][ky]
                    if self.half and kk in (7, 15, 31):
                        self.P[k + ".dw.tb"] = ops.toeplitz_bands(wd, self.dtype).to(self.device)     # MFMA B-operand bands
                    elif self.x3 and kk in (7, 15, 31) and os.environ.get("CFP_X3_DWLARGE", "1") != "0":
                        self.P[k + ".dw.tb"] = ops.toeplitz_bands_x3(wd).to(self.device)              # hi and lo band tables (f16x3)
                    self.P[k + ".dw.s"], self.P[k + ".dw.t"] = self._fold_bn(sd, k + ".bn1", sd[k + ".dwconv2.bias"], wd.shape[0], _BN_EPS)
                    self.P[k + ".norm.g"], self.P[k + ".norm.b"] = self._dev(sd[k + ".norm.weight"]), self._dev(sd[k + ".norm.bias"])
                    self._conv(sd, k + ".pw1", k + ".pwconv1.weight", k + ".pwconv1.bias")
                    self._conv(sd, k + ".pw2", k + ".pwconv2.weight", k + ".pwconv2.bias")
                else:
                    raise NotImplementedError(ln)
        h = "depth_head"
        # float32 master weights of decoder.conv0 in (kh, kw, c) order: the regressor branch gets mean(unet) from sums of conv0's INPUT
        w0 = sd[d + ".conv0.weight"].detach().float()
        self.P[d + ".conv0.w32"] = self._dev(w0.permute(0, 2, 3, 1).reshape(w0.shape[0], -1))
        self.P[d + ".conv0.b32"] = self._dev(sd[d + ".conv0.bias"])
        self._conv(sd, h + ".conv3x3", h + ".conv3x3.weight", h + ".conv3x3.bias")
        self.P[h + ".w1x1"] = self._dev(sd[h + ".conv1x1.weight"].reshape(128, 128).t())      # [in][out]
        for i in (0, 2, 4):
            self.P[f"{h}.r{i}.w"], self.P[f"{h}.r{i}.b"] = self._dev(sd[f"{h}.regressor.{i}.weight"].t()), self._dev(sd[f"{h}.regressor.{i}.bias"])
        self._conv(sd, "conv_out", "conv_out.0.weight", "conv_out.0.bias", w2=False)
        if self.half and self.n_bins == 256:
            # operand of the fused head kernel: K axis in the kernel's fragment order, hi + lo planes (ops.permute_wout)
            self.P["conv_out.wp"] = ops.permute_wout(sd["conv_out.0.weight"], self.dtype, hilo=self.head_hilo[0],
                                                   diffuse=self.diffuse_cin and self.dtype == torch.float16).to(self.device)

    # ------------------------------------------------------------------------------ buffers
    def _act(self, plan, key: str, rows: int, C: int, ld: Optional[int] = None, zero: bool = False) -> Act:
        bufs = plan["bufs"]
        if key not in bufs:
            bufs[key] = ops.new_act(rows, C, self.dtype, self.device, ld, zero)
        a = bufs[key]
        assert a.rows == rows and a.ld == (ld or C), key
        return Act(a.buf, 0, C)

    def _buf(self, plan, key: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        bufs = plan["bufs"]
        if key not in bufs:
            bufs[key] = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        assert tuple(bufs[key].shape) == tuple(shape), key
        return bufs[key]

    def _f32(self, plan, key: str, n: int) -> torch.Tensor:
        bufs = plan["bufs"]
        if key not in bufs:
            bufs[key] = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)
        assert bufs[key].numel() >= n, key
        return bufs[key]

    # ------------------------------------------------------------------------------ layers
    def _cv(self, name, x: Act, out: Act, B, H, W, k, stride=1, pads=None, act=hip.ACT_NONE, residual=None):
        if pads is None:
            p = (k - 1) // 2
            pt = pl = p
            Ho, Wo = (H + 2 * p - k) // stride + 1, (W + 2 * p - k) // stride + 1
        else:
            (pt, pb), (pl, pr) = pads
            Ho, Wo = (H + pt + pb - k) // stride + 1, (W + pl + pr - k) // stride + 1
        ops.conv2d(x, self.P[name + ".w"], self.P[name + ".s"], self.P[name + ".t"], out, B, H, W, k, k, stride, pt, pl,
                   Ho, Wo, act, residual, self._ws(B * Ho * Wo, out.C, k * k * x.C))
        return Ho, Wo

    def _lin(self, wname, x: Act, out: Act, rows, act=hip.ACT_NONE, residual=None, st: Optional[str] = None, ln=None):
        ops.linear(x, self.P[wname], self.P[st + ".s"] if st else None, self.P[st + ".t"] if st else None, out, rows, act, residual,
                   None if ln is not None else self._ws(rows, out.C, x.C), ln)

    def _ws(self, M: int, Cout: int, K: int) -> Optional[torch.Tensor]:
        """Split-K scratch shared by all layers (kernels on one stream run in order)."""
        need = ops.conv2d_ws_bytes(M, Cout, K, self.cdt)
        if need == 0:
            return None
        if self.tickets:
            need += hip.CONV_TICKET_BYTES
        ws = self._lane_ws.get(self._lane)
        if ws is None or ws.numel() * 4 < need:
            if self._capturing:
                raise RuntimeError("split-K workspace must be sized by a warm-up forward before graph capture")
            if self.tickets:      # f16x3: ticketed split-K (the last workgroup at a tile finishes it: no reduce launch)
                ws = ops.ticket_ws(need - hip.CONV_TICKET_BYTES, self.device)
            else:
                ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
            self._lane_ws[self._lane] = ws
        return ws

    def _dbg(self, name: str, a: Act):
        if self._dbg_round:
            for pat, dt in self._dbg_round.items():
                if name == pat or (pat.endswith("*") and name.startswith(pat[:-1])):
                    t = a.torch()
                    t.copy_(t.to(dt).to(t.dtype))

    def _encoder(self, plan, rgb: torch.Tensor, B, H, W, taps):
        """encoder.py:71-79 over timm tf_efficientnetv2_b3 blocks; returns the five tap Acts."""
        e = "img_encoder"
        x8 = self._act(plan, "rgb8", B * H * W, 8)
        (ops.rgb_to_nhwc8_hilo if self.rgb_hilo else ops.rgb_to_nhwc8)(rgb, x8, B, H, W)
        pads = (_same_pad(H, 3, 2), _same_pad(W, 3, 2))
        h, w = math.ceil(H / 2), math.ceil(W / 2)
        x = self._act(plan, "stem", B * h * w, spec.ENC_STEM_OUT)
        self._cv("stem", x8, x, B, H, W, 3, 2, pads, hip.ACT_SILU if self.stem_act else hip.ACT_NONE)
        self._dbg("stem", x)
        tap_acts: List[Act] = []
        for bi, b in enumerate(spec.ENC_BLOCKS):
            q = f"{e}.{b.prefix}"
            ho, wo = math.ceil(h / b.stride), math.ceil(w / b.stride)
            pads = (_same_pad(h, 3, b.stride), _same_pad(w, 3, b.stride))
            if bi in spec.ENC_TAPS:
                out = plan["tap_dst"][spec.ENC_TAPS[bi]]
            else:
                out = self._act(plan, f"enc{bi}", B * ho * wo, b.cout)
            res = x if b.skip else None
            if b.kind == "cn":
                self._cv(q + ".conv", x, out, B, h, w, 3, b.stride, pads, hip.ACT_SILU, res)
            elif b.kind == "er":
                mid = self._act(plan, f"enc{bi}.mid", B * ho * wo, b.mid)
                self._cv(q + ".exp", x, mid, B, h, w, 3, b.stride, pads, hip.ACT_SILU)
                self._dbg(f"enc{bi}.mid", mid)
                self._cv(q + ".pwl", mid, out, B, ho, wo, 1, 1, None, hip.ACT_NONE, res)
            else:
                mid2 = self._act(plan, f"enc{bi}.dw", B * ho * wo, b.mid)
                mbp = ops.mbconv_plan(B, h, w, x.C, b.mid) if (q + ".pw.wimg") in self.P else None
                if mbp is not None:
                    # expand GEMM + BN + SiLU + depthwise 3x3 + BN + SiLU + SE sums in one launch: the expanded tensor stays in LDS
                    ns = mbp[0]
                    part = self._f32(plan, f"enc{bi}.sum", B * ns * b.mid)
                    ops.mbconv_expand_dw(x, self.P[q + ".pw.wimg"], self.P[q + ".pw.s"], self.P[q + ".pw.t"], self.P[q + ".dw.w"],
                                         self.P[q + ".dw.s"], self.P[q + ".dw.t"], mid2, part, B, h, w)
                elif ((self.se2 or self.se2_x3 == "1" or (self.se2_x3 == "auto" and not ops.PLAN_IN_FLIGHT)) and b.se_rd <= 64
                      and ops.dwconv3x3_se_parts(B, ho, wo, b.mid, b.stride, ops.DT[self.dtype]) > 0):
                    # round 3: the depthwise kernel applies the reduce FC to its own channel sums (it is linear in them), the tail kernel
                    # adds the K partial vectors, finishes the gate and folds it into float32 project weights in one full-chip launch
                    mid = self._act(plan, f"enc{bi}.mid", B * h * w, b.mid)
                    self._cv(q + ".pw", x, mid, B, h, w, 1, 1, None, hip.ACT_SILU)
                    K = ops.dwconv3x3_se_parts(B, ho, wo, b.mid, b.stride, ops.DT[self.dtype])
                    hpart = self._f32(plan, f"enc{bi}.hpart", B * K * b.se_rd)
                    ops.dwconv3x3_se(mid, self.P[q + ".dw.w"], self.P[q + ".dw.s"], self.P[q + ".dw.t"], mid2, self.P[q + ".se.wr"], hpart,
                                     B, h, w, b.stride, pads[0][0], pads[1][0], ho, wo, hip.ACT_SILU)
                    if self.x3:      # per-image pre-split operands; the K padding of the rows is zeroed once and never written
                        wb = self._buf(plan, f"enc{bi}.wb", (B, b.cout, (b.mid + 31) // 32 * 64), torch.float16, zero=True)
                    else:
                        wb = self._buf(plan, f"enc{bi}.wb", (B, b.cout, b.mid), self.dtype)
                    ops.se_gate_fold2(hpart, K, 1.0 / (ho * wo), self.P[q + ".se.br"], self.P[q + ".se.we_t"], self.P[q + ".se.be"],
                                      self.P[q + ".pwl.w32"], wb, B, b.cout, b.mid, b.se_rd, x3=self.x3)
                    ops.conv2d(mid2, wb, self.P[q + ".pwl.s"], self.P[q + ".pwl.t"], out, B, ho, wo, 1, 1, 1, 0, 0, ho, wo, hip.ACT_NONE,
                               res, self._piw_ws(plan, bi, B, ho * wo, b.cout, b.mid), per_image_weights=True)
                    x, h, w = out, ho, wo
                    if bi in spec.ENC_TAPS:
                        tap_acts.append(out)
                        if taps is not None:
                            taps[f"enc{spec.ENC_TAPS[bi]}"] = self._nchw(out, B, h, w)
                    continue
                else:
                    mid = self._act(plan, f"enc{bi}.mid", B * h * w, b.mid)
                    self._cv(q + ".pw", x, mid, B, h, w, 1, 1, None, hip.ACT_SILU)
                    # depthwise + BN + SiLU, emitting the per-strip channel sums squeeze-excite needs
                    ns = ops.dwconv3x3_strips(B, ho, wo, b.mid, b.stride, ops.DT[self.dtype])
                    part = self._f32(plan, f"enc{bi}.sum", B * ns * b.mid)
                    self._dbg(f"enc{bi}.mid", mid)
                    ops.dwconv3x3_sum(mid, self.P[q + ".dw.w"], self.P[q + ".dw.s"], self.P[q + ".dw.t"], mid2, part, B, h, w, b.stride,
                                      pads[0][0], pads[1][0], ho, wo, hip.ACT_SILU)
                    self._dbg(f"enc{bi}.dw", mid2)
                # SE tail (mean -> FC -> SiLU -> FC -> sigmoid) in one launch; the gate multiplies the project conv's
                # input channels, so it is folded into per-image project weights instead of a pass over mid2
                if self.x3:      # per-image pre-split operands; the K padding of the rows is zeroed once and never written
                    wb = self._buf(plan, f"enc{bi}.wb", (B, b.cout, (b.mid + 31) // 32 * 64), torch.float16, zero=True)
                else:
                    wb = self._buf(plan, f"enc{bi}.wb", (B, b.cout, b.mid), self.dtype)
                ops.se_gate_fold(part, ns, 1.0 / (ho * wo), self.P[q + ".se.wr"], self.P[q + ".se.br"], self.P[q + ".se.we_t"],
                                 self.P[q + ".se.be"], self.P[q + (".pwl.w32" if self.x3 else ".pwl.w")], wb, B, b.cout, b.mid, b.se_rd)
                ops.conv2d(mid2, wb, self.P[q + ".pwl.s"], self.P[q + ".pwl.t"], out, B, ho, wo, 1, 1, 1, 0, 0, ho, wo, hip.ACT_NONE,
                           res, self._piw_ws(plan, bi, B, ho * wo, b.cout, b.mid), per_image_weights=True)
            self._dbg(f"enc{bi}", out)
            x, h, w = out, ho, wo
            if bi in spec.ENC_TAPS:
                tap_acts.append(out)
                if taps is not None:
                    taps[f"enc{spec.ENC_TAPS[bi]}"] = self._nchw(out, B, h, w)
        return tap_acts

    def _hist_encoder(self, plan, hist: torch.Tensor, R: int, taps):
        """encoder.py:45-50: nine pointwise layers in ONE launch, float32 arithmetic in every storage mode; only the three tapped
        embeddings are stored (in the storage type)."""
        outs = [self._act(plan, f"hist.{ex}", R, c) for ex, c in zip((1, 2, 3), spec.HIST_CHANNELS)]
        # every tap feeds exactly one fusion block, as `feat1 + positional_encodings2` (fusion.py:123-125): added on the way out
        # (tap ex -> cross_atten{ex}; with `taps` the raw embeddings are wanted, the engine then adds the table itself)
        self._hist_pe_fused = taps is None and os.environ.get("CFP_NO_PE_FUSE", "0") != "1"
        pe = [self.P[f"decoder.cross_atten{ex}.pe2"] for ex in (1, 2, 3)] if self._hist_pe_fused else (None, None, None)
        ops.hist_encoder(hist.reshape(-1), self.P["hist.blob"], self._hist_layout, outs, R, pe, self.zone_sample_num)
        if taps is not None:
            for ex, o in enumerate(outs):
                taps[f"hist{ex}"] = o.torch().float().cpu()
        return outs

    # -- LoFTR encoder layer (transformer.py:41-71) on tokens living in xb[:, 0:D] of a [rows, 2D] buffer
    def _kv_state(self, plan, tag, p, src: Act, rows_s, D, heads, kvmode: dict):
        """k|v projection of `src` + the linear-attention key/value state (KV, Ksum) of its groups -> (kv, ks).  For hist2image the
        source is the ToF embedding (fusion.py:123-125, 143): it depends on nothing the image path computes, so `forward` runs it
        for all six hist2image layers on the side stream beside the RGB encoder (18 launches off the critical path)."""
        d = D // heads
        kvb = self._act(plan, f"{tag}.kvsrc", rows_s, 2 * D)
        self._lin(p + ".kv", src, kvb, rows_s)
        G = kvmode["groups"]
        kv = self._f32(plan, f"{tag}.kv", G * heads * d * d)
        ks = self._f32(plan, f"{tag}.ks", G * heads * d)
        nws = ops.attn_kv_ws_floats(kvmode["NB"], kvmode["Hk"], kvmode["Wk"], kvmode["th"], kvmode["tw"], heads, d)
        ws = self._f32(plan, f"{tag}.kvws", nws)
        ops.attn_kv_reduce(kvb.slice(0, D), kvb.slice(D, D), kv, ks, ws, kvmode["NB"], kvmode["Hk"], kvmode["Wk"], kvmode["th"], kvmode["tw"],
                           kvmode["clip"], kvmode["count_pad"], kvmode["v_length"], heads, d)
        return kv, ks

    def _loftr(self, plan, tag, p, xb: Act, rows_q, src: Optional[Act], rows_s, heads, out: Act, kvmode: dict, apmode: dict, pre=None):
        D = xb.C // 2
        d = D // heads
        x = xb.slice(0, D)
        qb = self._act(plan, f"{tag}.qkv", rows_q, 3 * D)
        fused_tail = self.half or (self.x3 and os.environ.get("CFP_X3_TAIL", "1") != "0")      # apply + merge + norm1 + mlp + norm2 + residual in one kernel
        tail_q = fused_tail and self.tail_q       # the fused tail projects q for its own rows: no q GEMM, no q tensor
        if pre is not None and tail_q:
            kv, ks = pre                              # key/value state computed ahead of time (hist2image: _kv_state on the side stream)
            ops.loftr_tail(None, kv, ks, x, out, self.P[p + ".q1"], self.P[p + ".merge"], self.P[p + ".mlp0"], self.P[p + ".mlp2"],
                           (self.P[p + ".norm1.g"], self.P[p + ".norm1.b"]), (self.P[p + ".norm2.g"], self.P[p + ".norm2.b"]),
                           apmode["NB"], apmode["Hq"], apmode["Wq"], apmode["qth"], apmode["qtw"], kvmode["v_length"], heads)
            return
        if src is None and tail_q:
            self._lin(p + ".kv", x, qb.slice(D, 2 * D), rows_q)
            kA, vA = qb.slice(D, D), qb.slice(2 * D, D)
        elif src is None:     # self attention: one GEMM for q|k|v
            self._lin(p + ".qkv", x, qb, rows_q)
            kA, vA = qb.slice(D, D), qb.slice(2 * D, D)
        else:
            if not tail_q:
                self._lin(p + ".q", x, qb.slice(0, D), rows_q)
            kvb = self._act(plan, f"{tag}.kvsrc", rows_s, 2 * D)
            self._lin(p + ".kv", src, kvb, rows_s)
            kA, vA = kvb.slice(0, D), kvb.slice(D, D)
        G = kvmode["groups"]
        kv = self._f32(plan, f"{tag}.kv", G * heads * d * d)
        ks = self._f32(plan, f"{tag}.ks", G * heads * d)
        nws = ops.attn_kv_ws_floats(kvmode["NB"], kvmode["Hk"], kvmode["Wk"], kvmode["th"], kvmode["tw"], heads, d)
        ws = self._f32(plan, f"{tag}.kvws", nws)
        ops.attn_kv_reduce(kA, vA, kv, ks, ws, kvmode["NB"], kvmode["Hk"], kvmode["Wk"], kvmode["th"], kvmode["tw"],
                           kvmode["clip"], kvmode["count_pad"], kvmode["v_length"], heads, d)
        if fused_tail:
            # apply + merge + norm1 + mlp + norm2 + residual in one kernel: the intermediates stay in LDS
            ops.loftr_tail(None if tail_q else qb.slice(0, D), kv, ks, x, out, self.P[p + ".q1"] if tail_q else None,
                           self.P[p + ".merge"], self.P[p + ".mlp0"], self.P[p + ".mlp2"],
                           (self.P[p + ".norm1.g"], self.P[p + ".norm1.b"]), (self.P[p + ".norm2.g"], self.P[p + ".norm2.b"]),
                           apmode["NB"], apmode["Hq"], apmode["Wq"], apmode["qth"], apmode["qtw"], kvmode["v_length"], heads)
            return
        msg = self._act(plan, f"{tag}.msg", rows_q, D)
        ops.attn_apply(qb.slice(0, D), kv, ks, msg, apmode["NB"], apmode["Hq"], apmode["Wq"], apmode["qth"], apmode["qtw"],
                       (0, 0, 0, 0), kvmode["v_length"], heads, d)
        # merge -> norm1 and mlp -> norm2 -> + x: the LayerNorms run in the GEMM epilogues
        self._lin(p + ".merge", msg, xb.slice(D, D), rows_q, ln=(self.P[p + ".norm1.g"], self.P[p + ".norm1.b"], 1e-5))
        hid = self._act(plan, f"{tag}.hid", rows_q, 2 * D)
        self._lin(p + ".mlp0", xb, hid, rows_q, hip.ACT_RELU)
        self._lin(p + ".mlp2", hid, out, rows_q, residual=x, ln=(self.P[p + ".norm2.g"], self.P[p + ".norm2.b"], 1e-5))

    def _piw_ws(self, plan, bi: int, B: int, hw: int, cout: int, k: int):
        """Split-K slab workspace of a per-image-weight project GEMM in the f16x3 mode, when the kernel plan wants one (single images: a few
        row tiles x a long K; cfp_conv2d_plan with rows_per_batch = hw), else None."""
        if not self.x3:
            return None
        _, sp = ops.conv2d_plan(B * hw, cout, k, hip.F32X3, hw, B, 1, 1)
        if sp <= 1:
            return None
        if not self.tickets:
            return self._f32(plan, f"enc{bi}.piw_ws", sp * B * hw * cout)
        bufs, key = plan["bufs"], f"enc{bi}.piw_ws"
        if key not in bufs:
            bufs[key] = ops.ticket_ws(sp * B * hw * cout * 4, self.device)
        return bufs[key]

    def _fusion(self, plan, name: str, x: Act, feat1: Act, zone_valid: torch.Tensor, geo: FusionGeometry, B, H, W, out: Act,
                pos_offset, taps):
        """fusion.py:52-188."""
        D, (Hm, Wm), lk = self.fusion[name]
        p = f"decoder.{name}"
        ws = spec.window_size((Hm, Wm))
        M = B * H * W
        Z = geo.zone_num * geo.zone_num
        N = self.zone_sample_num
        tok = [self._act(plan, f"{name}.tokA", M, 2 * D), self._act(plan, f"{name}.tokB", M, 2 * D)]
        cur = 0
        if torch.is_tensor(pos_offset):
            # window origin read from DEVICE memory (int32[2]): a captured graph then serves every random window (fusion.py:87-91)
            assert pos_offset.dtype == torch.int32 and pos_offset.numel() == 2 and pos_offset.is_cuda
            d0 = tok[0].slice(0, D)
            hip.call("cfp_add_rowtable_dev", x.ptr, x.ld, self.P[p + ".pe"].data_ptr(), d0.ptr, d0.ld, M, x.C, H, W, Hm, Wm,
                     pos_offset.data_ptr(), x.dt, hip.current_stream())
        else:
            oy, ox = pos_offset
            ops.add_rowtable(x, self.P[p + ".pe"], tok[0].slice(0, D), M, H, W, Wm, oy, ox)
        emb0 = None
        if not self.change_embedding:
            emb0 = self._act(plan, f"{name}.emb0", M, D)
            ops.copy_rows(tok[0].slice(0, D), emb0, M)
        if self._hist_pe_fused:
            src = feat1                     # the ToF encoder kernel already added positional_encodings2
        else:
            src = self._act(plan, f"{name}.src", B * Z * N, D)
            ops.add_rowtable(feat1, self.P[p + ".pe2"], src, B * Z * N, 1, N, N, 0, 0)
        gh, gw = geo.grid_h, geo.grid_w
        Mz = B * gh * gw
        y0, y1, x0, x1 = geo.clipped(H, W)
        rect = (geo.sy_wo, geo.sx_wo, geo.tzh, geo.tzw)

        last = len(self.layer_names) - 1
        direct_out = False                  # the last layer's final kernel writes the block's output slice itself (no copy launch)
        for i, ln in enumerate(self.layer_names):
            l = f"{p}.layers.{i}"
            tag = f"{name}.L{i}"
            final_dst = out if (i == last and taps is None and os.environ.get("CFP_NO_DIRECT_OUT", "0") != "1") else None
            if ln == "hist2image":
                zin = self._act(plan, f"{name}.zin", Mz, 2 * D)
                zsrc = tok[cur].slice(0, D) if self.change_embedding else emb0
                ops.resize_bilinear(zsrc, H, W, rect, zin.slice(0, D), gh, gw, (0, 0, gh, gw), B)
                zout = self._act(plan, f"{name}.zout", Mz, D)
                self._loftr(plan, f"{name}.x2i", l, zin, Mz, src, B * Z * N, spec.X2I_HEADS, zout,
                            dict(groups=B * Z, NB=B * Z, Hk=1, Wk=N, th=1, tw=N, clip=(0, 1, 0, N), count_pad=False, v_length=float(N)),
                            dict(NB=B, Hq=gh, Wq=gw, qth=geo.p1, qtw=geo.p2), pre=plan.get("x2i_kv", {}).get((name, i)))
                ops.resize_bilinear(zout, gh, gw, (0, 0, gh, gw), tok[cur].slice(0, D), H, W, rect, B, zone_valid=zone_valid,
                                    zn=geo.zone_num, p1=geo.p1, p2=geo.p2, accumulate=not self.no_skip_inside)
            elif ln == "image":
                nh, nw = math.ceil(H / ws), math.ceil(W / ws)
                self._loftr(plan, f"{name}.lsa", l + ".lga.encoder_layer", tok[cur], M, None, 0, spec.TWINS_HEADS, tok[cur ^ 1].slice(0, D),
                            dict(groups=B * nh * nw, NB=B, Hk=H, Wk=W, th=ws, tw=ws, clip=(0, H, 0, W), count_pad=True, v_length=float(ws * ws)),
                            dict(NB=B, Hq=H, Wq=W, qth=ws, qtw=ws))
                cur ^= 1
                hk, wk = gsa_keys(H, W, ws)
                keys = self._act(plan, f"{name}.gsa.keys", B * hk * wk, D)
                if self.x3 and self.sr_ln_fused:
                    # f16x3: nn.LayerNorm goes into the patch conv -- into the finishing sum of its K splits, or the GEMM epilogue when K is not split
                    ops.conv2d(tok[cur].slice(0, D), self.P[l + ".gsa.sr.w"], self.P[l + ".gsa.sr.s"], self.P[l + ".gsa.sr.t"], keys,
                               B, H, W, ws, ws, ws, 0, 0, hk, wk, ws=self._ws(B * hk * wk, D, ws * ws * D),
                               ln=(self.P[l + ".gsa.norm.g"], self.P[l + ".gsa.norm.b"], 1e-5))
                else:
                    kraw = self._act(plan, f"{name}.gsa.kraw", B * hk * wk, D)
                    ops.conv2d(tok[cur].slice(0, D), self.P[l + ".gsa.sr.w"], self.P[l + ".gsa.sr.s"], self.P[l + ".gsa.sr.t"], kraw,
                               B, H, W, ws, ws, ws, 0, 0, hk, wk, ws=self._ws(B * hk * wk, D, ws * ws * D))
                    ops.layernorm(kraw, self.P[l + ".gsa.norm.g"], self.P[l + ".gsa.norm.b"], 1e-5, keys, B * hk * wk)
                S = hk * wk
                self._loftr(plan, f"{name}.gsa", l + ".gsa.encoder_layer", tok[cur], M, keys, B * S, spec.TWINS_HEADS,
                            final_dst if final_dst is not None else tok[cur ^ 1].slice(0, D),
                            dict(groups=B, NB=B, Hk=1, Wk=S, th=1, tw=S, clip=(0, 1, 0, S), count_pad=False, v_length=float(S)),
                            dict(NB=B, Hq=H, Wq=W, qth=H, qtw=W))
                cur ^= 1
                direct_out = final_dst is not None
            elif ln == "combine1":
                # DAPM (transformer.py:204-248)
                t = l + ".transformer_path"
                heads = spec.X2I_HEADS
                d = D // heads
                xin = tok[cur]
                qb = self._act(plan, f"{name}.dapm.qkv", M, 3 * D)
                self._lin(t + ".qkv", xin.slice(0, D), qb, M)
                n_in = (y1 - y0) * (x1 - x0)
                kv = self._f32(plan, f"{name}.dapm.kv", B * heads * d * d)
                ks = self._f32(plan, f"{name}.dapm.ks", B * heads * d)
                wsb = self._f32(plan, f"{name}.dapm.ws", ops.attn_kv_ws_floats(B, H, W, H, W, heads, d))
                ops.attn_kv_reduce(qb.slice(D, D), qb.slice(2 * D, D), kv, ks, wsb, B, H, W, H, W, (y0, y1, x0, x1), False,
                                   float(max(n_in, 1)), heads, d)
                ops.attn_apply(qb.slice(0, D), kv, ks, xin.slice(D, D), B, H, W, H, W, (y0, y1, x0, x1), float(max(n_in, 1)), heads, d)
                c1 = self._act(plan, f"{name}.dapm.c1", M, D)
                self._cv(t + ".conv1", xin, c1, B, H, W, 3)
                self._cv(t + ".conv2", c1, tok[cur ^ 1].slice(0, D), B, H, W, 3, residual=xin.slice(0, D))
                cur ^= 1
                # LKPM (convnext.py:42-58)
                k = l + ".large_kernel_path"
                xin = tok[cur]
                t1 = self._act(plan, f"{name}.lk.t1", M, D)
                if (k + ".dw.tb") in self.P:
                    ops.dwconv_large_mfma(xin.slice(0, D), self.P[k + ".dw.tb"], self.P[k + ".dw.s"], self.P[k + ".dw.t"], t1, B, H, W, lk,
                                          hip.ACT_RELU)
                else:
                    ops.dwconv_large(xin.slice(0, D), self.P[k + ".dw.w"], self.P[k + ".dw.s"], self.P[k + ".dw.t"], t1, B, H, W, lk, hip.ACT_RELU)
                lk_dst = final_dst if final_dst is not None else tok[cur ^ 1].slice(0, D)
                if ((self.half and not self.weights2) or (self.x3 and os.environ.get("CFP_X3_TAIL", "1") != "0")) and self.lkpm_fused and D in (32, 64, 128) and M >= self.lkpm_min_rows:
                    # LayerNorm -> pwconv1 -> GELU -> pwconv2 -> + input in one kernel: the 4D-wide hidden tensor stays in LDS.
                    # Measured at batch 8 (tools/small_kernel_bench.py): 24 vs 49 us at the 1/4 scale, 29 vs 34 us at 1/8, 30 vs 25 us at
                    # 1/16 (9 600 rows = 150 workgroups: too few to hide the chain's latency) -> only the many-row scales take it
                    ops.lkpm_tail(t1, xin.slice(0, D), lk_dst, self.P[k + ".pw1.w"], self.P[k + ".pw1.t"], self.P[k + ".pw2.w"], self.P[k + ".pw2.t"],
                                  self.P[k + ".norm.g"], self.P[k + ".norm.b"], M)
                else:
                    t2 = self._act(plan, f"{name}.lk.t2", M, D)
                    ops.layernorm(t1, self.P[k + ".norm.g"], self.P[k + ".norm.b"], 1e-6, t2, M)
                    h4 = self._act(plan, f"{name}.lk.h4", M, 4 * D)
                    self._lin(k + ".pw1.w", t2, h4, M, hip.ACT_GELU, None, k + ".pw1")
                    self._lin(k + ".pw2.w", h4, lk_dst, M, hip.ACT_NONE, xin.slice(0, D), k + ".pw2")
                cur ^= 1
                direct_out = final_dst is not None
            else:
                raise NotImplementedError(ln)
            if taps is not None:
                taps[f"{p}.layers.{i}"] = tok[cur].slice(0, D).torch().float().cpu().reshape(B, H * W, D)
        if not direct_out:
            ops.copy_rows(tok[cur].slice(0, D), out, M)

    def _nchw(self, a: Act, B, H, W) -> torch.Tensor:
        return a.torch().float().cpu().reshape(B, H, W, a.C).permute(0, 3, 1, 2).contiguous()

    # ------------------------------------------------------------------------------ batch lanes
    @torch.no_grad()
    def forward_lanes(self, input_data: dict, lanes: int, *, return_prob: bool = True, pos_offsets: Optional[dict] = None):
        """The same forward with the batch cut into `lanes` contiguous sub-batches that run CONCURRENTLY, each on its own
        HIP stream with its own scratch buffers, writing into slices of one set of outputs.  At batch 8 a forward is ~280
        dependent kernels that each leave most of the 256 CUs idle while they wait on their own loads (a batch-1 forward
        takes 3.1 ms, a batch-8 one 5.2 ms): independent sample groups fill those gaps.  The zone geometry stays the
        batch-reduced geometry of the WHOLE batch (fusion.py:75-84), so results are those of the unsplit forward."""
        rgb = input_data["rgb"]
        B = rgb.shape[0]
        lanes = max(1, min(lanes, B))
        if lanes == 1:
            return self.forward(input_data, return_prob=return_prob, pos_offsets=pos_offsets)
        dev = self.device
        add = input_data["additional"]
        H, W = rgb.shape[-2:]
        edges = torch.empty(B, self.n_bins + 1, dtype=torch.float32, device=dev)
        pred = torch.empty(B, 1, H // 2, W // 2, dtype=torch.float32, device=dev)
        prob = torch.empty(B, self.n_bins, H // 2, W // 2, dtype=self.dtype, device=dev) if return_prob else None
        main = torch.cuda.current_stream(dev)
        bounds = [B * i // lanes for i in range(lanes + 1)]
        for i in range(lanes):
            if i not in self._lane_streams and i > 0:
                self._lane_streams[i] = (_shared_stream(dev, ("lane", i)), _shared_stream(dev, ("lane-side", i)))
        rgb_d = rgb.to(device=dev, dtype=torch.float32).contiguous()
        hist_d = add["hist_data"].to(device=dev, dtype=torch.float32).contiguous()
        mask_d = add["mask"].to(device=dev).to(torch.uint8).contiguous()   # converted once: lanes must not allocate while capturing
        self._lanes_active = lanes
        for i in range(lanes):
            b0, b1 = bounds[i], bounds[i + 1]
            sub = {"rgb": rgb_d[b0:b1], "additional": {"hist_data": hist_d[b0:b1], "mask": mask_d[b0:b1],
                                                       "rect_data": add.get("rect_data"), "patch_info": add["patch_info"]}}
            outs = (edges[b0:b1], pred[b0:b1], prob[b0:b1] if prob is not None else None)
            if i == 0:
                self.forward(sub, return_prob=return_prob, pos_offsets=pos_offsets, lane=0, out=outs)
            else:
                st = self._lane_streams[i][0]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    self.forward(sub, return_prob=return_prob, pos_offsets=pos_offsets, lane=i, out=outs)
        for i in range(1, lanes):
            main.wait_stream(self._lane_streams[i][0])
        self._lane = 0
        self._lanes_active = 1
        return edges, pred, prob

    # ------------------------------------------------------------------------------ HIP graph
    def capture(self, input_data: dict, *, return_prob: bool = True, pos_offsets: Optional[dict] = None, lanes: int = 1,
                inflight: int = 1, adopt_inputs: bool = False):
        """Record the whole forward for this input shape into HIP graphs.  The launch list is static (all
        data-dependent geometry is host-side integers), so replaying costs one graph launch instead of ~280 kernel
        launches.  With `lanes` > 1 every batch lane gets its OWN graph, captured on and replayed from its own stream
        (one multi-stream capture of several 280-node branches crashes hipStreamEndCapture on ROCm 7.2); the lane
        graphs run concurrently and write into slices of one set of output tensors.

        `inflight` > 1 instead keeps that many WHOLE batches in flight: one linear graph per slot, each with its own
        inputs, scratch and outputs, on streams probed to run concurrently (`concurrent_streams`).  Use `replay_async`
        to feed the slots round-robin; `replay` still gives the plain one-call-one-result behaviour.

        `adopt_inputs` (single graph only): the graph reads the CALLER'S device tensors in place instead of private copies (they must be
        float32 / uint8-or-bool, contiguous, on this device and must stay alive): `replay()` without arguments then sees whatever they
        hold at that moment, with no copy in front of it (`Deltar.forward` uses this when it is called with the same tensors again)."""
        dev = self.device
        add = input_data["additional"]
        B = input_data["rgb"].shape[0]
        H, W = input_data["rgb"].shape[-2:]
        lanes = max(1, min(lanes, B))
        self._slots = None
        self._slot_next = 0
        if inflight > 1:
            return self._capture_inflight(input_data, inflight, return_prob, pos_offsets)
        if adopt_inputs:
            assert lanes == 1, "adopt_inputs: single-graph capture only"
            m = add["mask"]
            for t, dts in ((input_data["rgb"], (torch.float32,)), (add["hist_data"], (torch.float32,)), (m, (torch.uint8, torch.bool))):
                assert t.device == dev and t.dtype in dts and t.is_contiguous(), "adopt_inputs: device tensors in the kernels' own types"
            static = {"rgb": input_data["rgb"], "additional": {"hist_data": add["hist_data"], "mask": m.view(torch.uint8) if m.dtype == torch.bool else m,
                                                               "rect_data": add.get("rect_data"), "patch_info": add["patch_info"]}}
        else:
            static = {"rgb": input_data["rgb"].to(device=dev, dtype=torch.float32).contiguous().clone(),
                      "additional": {"hist_data": add["hist_data"].to(device=dev, dtype=torch.float32).contiguous().clone(),
                                     "mask": add["mask"].to(device=dev).to(torch.uint8).contiguous().clone(),
                                     "rect_data": add.get("rect_data"), "patch_info": add["patch_info"]}}
        edges = torch.empty(B, self.n_bins + 1, dtype=torch.float32, device=dev)
        pred = torch.empty(B, 1, H // 2, W // 2, dtype=torch.float32, device=dev)
        prob = torch.empty(B, self.n_bins, H // 2, W // 2, dtype=self.dtype, device=dev) if return_prob else None
        bounds = [B * i // lanes for i in range(lanes + 1)]
        subs, outs = [], []
        for i in range(lanes):
            b0, b1 = bounds[i], bounds[i + 1]
            sa = static["additional"]
            subs.append({"rgb": static["rgb"][b0:b1], "additional": {"hist_data": sa["hist_data"][b0:b1], "mask": sa["mask"][b0:b1],
                                                                     "rect_data": sa["rect_data"], "patch_info": sa["patch_info"]}})
            outs.append((edges[b0:b1], pred[b0:b1], prob[b0:b1] if prob is not None else None))
            if i not in self._lane_streams:
                self._lane_streams[i] = (_shared_stream(dev, ("lane", i)), _shared_stream(dev, ("lane-side", i)))
        graphs = []
        self._graph = None                      # drop the previous graphs before instantiating new ones
        self._lanes_active = lanes
        cur = torch.cuda.current_stream(dev)
        for i in range(lanes):
            st = self._lane_streams[i][0]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for _ in range(2):      # warm-up: allocates every buffer of the lane's plan, sets kernel attributes
                    self.forward(subs[i], return_prob=return_prob, pos_offsets=pos_offsets, lane=i, out=outs[i])
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            self._capturing = True
            try:
                with torch.cuda.graph(g, stream=st):
                    self.forward(subs[i], return_prob=return_prob, pos_offsets=pos_offsets, lane=i, out=outs[i])
            finally:
                self._capturing = False
            graphs.append(g)
        torch.cuda.synchronize(dev)
        self._lane = 0
        self._lanes_active = 1
        self._graph = (graphs, static, (edges, pred, prob))
        return edges, pred, prob

    def plan_mode(self, throughput: bool):
        """Which of the two kernel plans the convolutions launched from now on follow: the default one, fitted on isolated timings, or the
        one for several batches in flight (larger tiles: a launch's cost is then the resources it holds, not its own latency; the
        CFP_CONV_IN_FLIGHT hint of cfp_conv2d_nhwc_ex).  `capture(inflight=n)` switches to the second while it records its slots.  Results
        are bit-identical WITHIN a plan; across the two plans a convolution may run through another kernel (halo / implicit GEMM, another
        tile), i.e. another float32 summation order -- re-association noise, far below the storage rounding of the 16-bit modes."""
        ops.PLAN_IN_FLIGHT = bool(throughput) and os.environ.get("CFP_TPUT_PLAN", "1") != "0"

    def _capture_inflight(self, input_data, inflight, return_prob, pos_offsets):
        self.plan_mode(True)
        try:
            return self._capture_inflight_impl(input_data, inflight, return_prob, pos_offsets)
        finally:
            self.plan_mode(False)

    def _capture_inflight_impl(self, input_data, inflight, return_prob, pos_offsets):
        dev = self.device
        add = input_data["additional"]
        B = input_data["rgb"].shape[0]
        H, W = input_data["rgb"].shape[-2:]
        streams = concurrent_streams(dev, want=inflight)
        self._graph = None
        self._lanes_active = max(2, len(streams))          # linear graphs: no in-graph side-stream fork
        cur = torch.cuda.current_stream(dev)
        slots = []
        for si, st in enumerate(streams):
            static = {"rgb": input_data["rgb"].to(device=dev, dtype=torch.float32).contiguous().clone(),
                      "additional": {"hist_data": add["hist_data"].to(device=dev, dtype=torch.float32).contiguous().clone(),
                                     "mask": add["mask"].to(device=dev).to(torch.uint8).contiguous().clone(),
                                     "rect_data": add.get("rect_data"), "patch_info": add["patch_info"]}}
            out = (torch.empty(B, self.n_bins + 1, dtype=torch.float32, device=dev),
                   torch.empty(B, 1, H // 2, W // 2, dtype=torch.float32, device=dev),
                   torch.empty(B, self.n_bins, H // 2, W // 2, dtype=self.dtype, device=dev) if return_prob else None)
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for _ in range(2):
                    self.forward(static, return_prob=return_prob, pos_offsets=pos_offsets, lane=si, out=out)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            self._capturing = True
            try:
                with torch.cuda.graph(g, stream=st):
                    self.forward(static, return_prob=return_prob, pos_offsets=pos_offsets, lane=si, out=out)
            finally:
                self._capturing = False
            slots.append({"graph": g, "static": static, "out": out, "stream": st, "event": torch.cuda.Event()})
        torch.cuda.synchronize(dev)
        self._lane = 0
        self._lanes_active = 1
        self._slots = slots
        return slots[0]["out"]

    def replay_async(self, input_data: Optional[dict] = None):
        """Launch the captured forward on the next in-flight slot WITHOUT making the current stream wait for it.
        Returns ((edges, pred, prob), event): the tensors belong to the slot and are valid from `event` until the slot
        comes round again (`inflight` calls later).  Inputs are read on the slot's stream after the current stream's
        pending work.  Falls back to `replay` (+ an event on the current stream) when captured without `inflight`."""
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        if not self._slots:
            out = self.replay(input_data)
            ev = torch.cuda.Event()
            ev.record(cur)
            return out, ev
        slot = self._slots[self._slot_next % len(self._slots)]
        self._slot_next += 1
        st = slot["stream"]
        with torch.cuda.stream(st):
            if input_data is not None:
                st.wait_stream(cur)
                static = slot["static"]
                static["rgb"].copy_(input_data["rgb"], non_blocking=True)
                static["additional"]["hist_data"].copy_(input_data["additional"]["hist_data"], non_blocking=True)
                static["additional"]["mask"].copy_(input_data["additional"]["mask"].to(torch.uint8), non_blocking=True)
            slot["graph"].replay()
            slot["event"].record(st)
        return slot["out"], slot["event"]

    def capture_best(self, input_data: dict, *, return_prob: bool = True, pos_offsets: Optional[dict] = None,
                     candidates: Sequence = (("lanes", 1), ("inflight", 2), ("inflight", 3), ("inflight", 4), ("inflight", 6)),
                     reps: int = 16, allow_inflight: bool = True):
        """capture() with the concurrency mode chosen by measurement: `lanes` sub-batches of one batch side by side, or
        `inflight` whole batches in flight (throughput mode, results through `replay_async`).  How well graphs overlap
        depends on how the runtime maps streams onto its hardware queues and on what else in the process owns streams
        (RCCL, a profiler), so the choice is timed, not assumed.  Returns ((kind, n), {"kind:n": ms_per_step})."""
        import time
        B = input_data["rgb"].shape[0]
        times = {}
        seen = []
        for kind, n in candidates:
            n = max(1, int(n))
            if kind == "lanes":
                n = min(n, B)
            elif not allow_inflight:
                continue
            if (kind, n) in seen:
                continue
            seen.append((kind, n))
            last = f"{kind}:{n}"
            if kind == "lanes":
                self.capture(input_data, return_prob=return_prob, pos_offsets=pos_offsets, lanes=n)
                run = self.replay
            else:
                self.capture(input_data, return_prob=return_prob, pos_offsets=pos_offsets, inflight=n)
                if len(self._slots) < n:           # fewer concurrent streams than asked for: same as a smaller n
                    continue
                run = self.replay_async
            for _ in range(2 * n + 2):
                run()
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize(self.device)
            times[f"{kind}:{n}"] = (time.perf_counter() - t0) / reps * 1e3
        best = min(times, key=times.get)
        kind, n = best.split(":")
        n = int(n)
        if best != last:
            if kind == "lanes":
                self.capture(input_data, return_prob=return_prob, pos_offsets=pos_offsets, lanes=n)
            else:
                self.capture(input_data, return_prob=return_prob, pos_offsets=pos_offsets, inflight=n)
        return (kind, n), times

    def replay(self, input_data: Optional[dict] = None):
        """Re-run the captured forward; `input_data` (same shapes) is copied into the static inputs."""
        if self._slots:
            out, ev = self.replay_async(input_data)
            torch.cuda.current_stream(self.device).wait_event(ev)
            return out
        if self._graph is None:
            raise RuntimeError("Engine.replay() before capture()")
        graphs, static, out = self._graph
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        if input_data is not None:
            static["rgb"].copy_(input_data["rgb"], non_blocking=True)
            static["additional"]["hist_data"].copy_(input_data["additional"]["hist_data"], non_blocking=True)
            m = input_data["additional"]["mask"]
            if m.dtype == torch.bool and m.is_cuda:      # same bytes as uint8: a plain device copy, no conversion kernel
                static["additional"]["mask"].view(torch.bool).copy_(m, non_blocking=True)
            else:
                static["additional"]["mask"].copy_(m.to(torch.uint8), non_blocking=True)
        if len(graphs) == 1:
            st = self._lane_streams[0][0]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                graphs[0].replay()
            cur.wait_stream(st)
            return out
        for i, g in enumerate(graphs):
            st = self._lane_streams[i][0]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                g.replay()
        for i in range(len(graphs)):
            cur.wait_stream(self._lane_streams[i][0])
        return out

    # ------------------------------------------------------------------------------ forward
    def _plan(self, B, H, W) -> dict:
        key = (B, H, W, self._lane)
        if key not in self._plans:
            self._plans[key] = {"bufs": {}}
        return self._plans[key]

    @torch.no_grad()
    def forward(self, input_data: dict, *, return_prob: bool = True, pos_offsets: Optional[dict] = None,
                taps: Optional[dict] = None, img_features: Optional[Sequence[torch.Tensor]] = None, lane: int = 0,
                out: Optional[tuple] = None):
        """Eval-mode forward.  Returns (bin_edges [B,n+1] f32, pred [B,1,H/2,W/2] f32, prob [B,n,H/2,W/2] | None).
        `lane` selects an independent set of scratch buffers / side stream (forward_lanes); `out` = preallocated
        (edges, pred, prob) views to write into."""
        self._lane = lane
        side = self._lane_streams[lane][1] if lane in self._lane_streams else self._side
        if not self.use_side_stream or self._lanes_active > 1:
            # several batch lanes already overlap each other; keeping every lane's graph a LINEAR chain also keeps the
            # runtime from spreading it over internal streams -- a re-captured 2-lane graph with in-graph forks measured
            # 7.3 ms/step (lanes serialised on one hardware queue) against 4.5 ms for linear lane graphs
            side = torch.cuda.current_stream(self.device)
        rgb = input_data["rgb"]
        add = input_data["additional"]
        B, _, H, W = rgb.shape
        assert H % 32 == 0 and W % 32 == 0, "input height/width must be multiples of 32"
        dev = self.device
        plan = self._plan(B, H, W)
        pos_offsets = pos_offsets or {}
        hs = [H // 2, H // 4, H // 8, H // 16, H // 32]
        wsz = [W // 2, W // 4, W // 8, W // 16, W // 32]
        e, c = spec.DEC_ENC_CH, spec.DEC_CH
        # concatenation buffers: [upsampled decoder features | encoder skip]
        cp = (lambda n: (n + 31) // 32 * 32) if (self.x3 and self.cat_pad) else (lambda n: n)      # f16x3: zero channels up to a multiple of 32, never written
        zc = self.x3 and self.cat_pad
        cat = [None,
               self._act(plan, "cat1", B * hs[3] * wsz[3], cp(c[0] + e[1]), zero=zc),
               self._act(plan, "cat2", B * hs[2] * wsz[2], cp(c[1] + e[2]), zero=zc),
               self._act(plan, "cat3", B * hs[1] * wsz[1], cp(c[2] + e[3]), zero=zc),
               self._act(plan, "cat4", B * hs[0] * wsz[0], cp(c[3] + e[4]), zero=zc)]
        b4 = self._act(plan, "tap4", B * hs[4] * wsz[4], e[0])
        plan["tap_dst"] = [cat[4].slice(c[3], e[4]), cat[3].slice(c[2], e[3]), cat[2].slice(c[1], e[2]),
                           cat[1].slice(c[0], e[1]), b4]
        main = torch.cuda.current_stream(dev)
        hist = add["hist_data"].to(device=dev, dtype=torch.float32).contiguous()
        zone_valid = add["mask"].to(device=dev).to(torch.uint8).contiguous()
        Z, N = hist.shape[1], hist.shape[2]
        side.wait_stream(main)
        with torch.cuda.stream(side):                # ToF branch: 10 tiny launches, hidden under the RGB encoder
            hfeat = self._hist_encoder(plan, hist, B * Z * N, taps)
            plan["x2i_kv"] = {}
            if (self.half or (self.x3 and os.environ.get("CFP_X3_TAIL", "1") != "0")) and self.tail_q and self._hist_pe_fused and os.environ.get("CFP_X2I_HOIST", "1") == "1":
                # the key/value states of all hist2image layers depend on the ToF embeddings only: beside the RGB encoder too
                for fname, feat in (("cross_atten3", hfeat[2]), ("cross_atten2", hfeat[1]), ("cross_atten1", hfeat[0])):
                    D = self.fusion[fname][0]
                    for i, ln in enumerate(self.layer_names):
                        if ln == "hist2image":
                            plan["x2i_kv"][(fname, i)] = self._kv_state(
                                plan, f"{fname}.x2i{i}", f"decoder.{fname}.layers.{i}", feat, B * Z * N, D, spec.X2I_HEADS,
                                dict(groups=B * Z, NB=B * Z, Hk=1, Wk=N, th=1, tw=N, clip=(0, 1, 0, N), count_pad=False, v_length=float(N)))
        if img_features is not None:      # test hook: bypass the RGB encoder with given NCHW features
            for f, dst in zip(img_features, plan["tap_dst"]):
                dst.torch().copy_(f.permute(0, 2, 3, 1).reshape(-1, f.shape[1]).to(device=dev, dtype=self.dtype))
        else:
            self._encoder(plan, rgb.to(device=dev, dtype=torch.float32).contiguous(), B, H, W, taps)

        pinfo = add["patch_info"]

        def fuse(name, x, feat, hh, ww, out):
            Wm = self.fusion[name][1][1]
            geo = FusionGeometry.from_patch_info(pinfo, self.base_resolution[1] / Wm)      # fusion.py:41 (the stride: 16 / 8 / 4)
            assert geo.zone_num * geo.zone_num == Z
            self._fusion(plan, name, x, feat, zone_valid, geo, B, hh, ww, out, pos_offsets.get(name, (0, 0)), taps)

        def up(i, src: Act, hs_, ws_, hd, wd):
            M = B * hd * wd
            t1 = self._act(plan, f"up{i}.a", M, c[i])
            # the fused kernel's own preconditions (cfp_upsample_cat_conv3x3 returns CFP_ESHAPE otherwise): 32-bit byte offsets, H, W > 1
            fits = hd > 1 and wd > 1 and B * hd * wd * cat[i].ld * 2 < 2 ** 31 - 65536 and hs_ * ws_ * src.ld * 2 < 2 ** 31 - 65536
            if self.x3 and str(i) in self.up_fused_x3 and src.C % 32 == 0 and taps is None and hd > 1 and wd > 1:
                # round 5: the two-source chunk kernel (conv3x3_halo_x3.hip, UP): bilinear blend in the halo loader, skip channels from the
                # concatenation buffer's slice; the upsampled tensor is never written
                n = f"decoder.up{i}.a"
                ops.upsample_cat_conv3x3(src, hs_, ws_, cat[i].slice(src.C, e[i]), self.P[n + ".wcat"], self.P[n + ".s"], self.P[n + ".t"],
                                         t1, B, hd, wd, hip.ACT_LRELU, x3=True)
            elif self.half and str(i) in self.up_fused and src.C % 64 == 0 and taps is None and fits:
                # cfp_upsample_cat_conv3x3: bilinear + concat computed inside the conv's halo loader (bit-identical to the pair below)
                n = f"decoder.up{i}.a"
                ops.upsample_cat_conv3x3(src, hs_, ws_, cat[i].slice(src.C, e[i]), self.P[n + ".w"], self.P[n + ".s"], self.P[n + ".t"],
                                         t1, B, hd, wd, hip.ACT_LRELU)
            else:
                ops.resize_bilinear(src, hs_, ws_, (0, 0, hs_, ws_), cat[i].slice(0, src.C), hd, wd, (0, 0, hd, wd), B)
                n = f"decoder.up{i}.a"
                if (n + ".wpad") in self.P and M >= 30000:
                    # many pixels: the zero-padded concatenation (Cin % 32 == 0) through the chunk-pipelined kernel; few pixels (single images
                    # below the 1/2 scale): the unpadded columns of the same buffer through the implicit GEMM as before
                    ops.conv2d(cat[i], self.P[n + ".wpad"], self.P[n + ".s"], self.P[n + ".t"], t1, B, hd, wd, 3, 3, 1, 1, 1, hd, wd, hip.ACT_LRELU,
                               None, self._ws(M, t1.C, 9 * cat[i].C))
                else:
                    self._cv(n, cat[i].slice(0, src.C + e[i]), t1, B, hd, wd, 3, act=hip.ACT_LRELU)
            t2 = self._act(plan, f"up{i}.b", M, c[i])
            self._cv(f"decoder.up{i}.b", t1, t2, B, hd, wd, 3, act=hip.ACT_LRELU)
            return t2

        main.wait_stream(side)                        # join: the decoder consumes the ToF embeddings
        xd4 = self._act(plan, "xd4", B * hs[4] * wsz[4], c[0])
        self._cv("decoder.conv4", b4, xd4, B, hs[4], wsz[4], 1)
        x = xd4
        ph, pw = hs[4], wsz[4]
        for i, (fname, feat) in enumerate((("cross_atten3", hfeat[2]), ("cross_atten2", hfeat[1]), ("cross_atten1", hfeat[0])), start=1):
            hh, ww = hs[4 - i], wsz[4 - i]
            t = up(i, x, ph, pw, hh, ww)
            if taps is not None:
                taps[f"up{i}"] = self._nchw(t, B, hh, ww)
            D = c[i + 1]
            dcat = self._act(plan, f"dcat{i}", B * hh * ww, 2 * D)
            self._cv(f"decoder.conv{4 - i}", t, dcat.slice(0, D), B, hh, ww, 1)
            fuse(fname, dcat.slice(0, D), feat, hh, ww, dcat.slice(D, D))
            if taps is not None:
                taps[f"conv{4 - i}"] = self._nchw(dcat.slice(0, D), B, hh, ww)
                taps[fname] = self._nchw(dcat.slice(D, D), B, hh, ww)
            x, ph, pw = dcat, hh, ww
        t = up(4, x, ph, pw, hs[0], wsz[0])
        Mh = B * hs[0] * wsz[0]
        HWh = hs[0] * wsz[0]
        unet = self._act(plan, "unet", Mh, 128)
        fused_head = self.half and self.head_fused and self.n_bins == 256 and HWh % 16 == 0 and HWh >= 128 and "conv_out.wp" in self.P
        ram = self._act(plan, "ram", Mh, 128) if (not fused_head or taps is not None) else None
        edges = out[0] if out is not None else torch.empty(B, self.n_bins + 1, dtype=torch.float32, device=dev)
        centers = self._f32(plan, "head.centers", B * self.n_bins)
        h = "depth_head"
        # bin-width regressor branch (decoder.py:28-36).  Its input mean_HW(conv1x1(unet)) is linear in unet = conv0(t), a linear
        # 3x3 conv: the spatial sum of unet follows from nine shifted sums of the 32-channel t (cfp_channel_sum + cfp_conv3x3_mean),
        # so the whole branch depends on t only and runs BESIDE conv0 on the side stream -- off the critical path conv0 -> head.
        ns = max(1, min(64, HWh // 256))
        bsum = self._f32(plan, "head.bsum", B * ns * t.C)
        msum = self._f32(plan, "head.msum", B * 128)
        if os.environ.get("CFP_OLD_SUMS", "0") == "1":      # probe switch: the round-1 order (sums of unet after conv0)
            self._cv("decoder.conv0", t, unet, B, hs[0], wsz[0], 3)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.channel_sum(t, bsum, B, HWh, ns)
            ops.conv3x3_mean(bsum, ns, t, self.P["decoder.conv0.w32"], self.P["decoder.conv0.b32"], msum, B, hs[0], wsz[0], 128)
            ops.bin_regressor(msum, 1, 1.0 / HWh, self.P[h + ".w1x1"], self.P[h + ".r0.w"], self.P[h + ".r0.b"], self.P[h + ".r2.w"],
                              self.P[h + ".r2.b"], self.P[h + ".r4.w"], self.P[h + ".r4.b"], self.min_val, self.max_val, self.norm,
                              edges, centers, B, 128, 256, self.n_bins)
        if os.environ.get("CFP_OLD_SUMS", "0") != "1":
            self._cv("decoder.conv0", t, unet, B, hs[0], wsz[0], 3)
        if not fused_head:
            self._cv("depth_head.conv3x3", unet, ram, B, hs[0], wsz[0], 3)
        main.wait_stream(side)
        pred = out[1] if out is not None else torch.empty(B, 1, hs[0], wsz[0], dtype=torch.float32, device=dev)
        if out is not None:
            prob = out[2] if return_prob else None
        else:
            prob = torch.empty(B, self.n_bins, hs[0], wsz[0], dtype=self.dtype, device=dev) if return_prob else None
        if fused_head:
            # conv3x3 + conv_out + softmax + expectation in one kernel: neither ram nor the logits reach HBM
            ops.depth_head_fused(unet, self.P[h + ".conv3x3.w"], self.P[h + ".conv3x3.s"], self.P[h + ".conv3x3.t"], self.P["conv_out.wp"],
                                 self.P["conv_out.t"], centers, prob, pred, B, hs[0], wsz[0], ram_out=ram, ram_hilo=self.head_hilo[1])
        elif (self.half or self.x3) and self.n_bins == 256 and HWh % 8 == 0 and os.environ.get("CFP_BIN_HEAD_FUSED", "1") != "0":
            # 1x1 conv + softmax + expectation in one kernel: the logits never reach HBM
            ops.bin_head_fused(ram, self.P["conv_out.w"], self.P["conv_out.t"], centers, prob, pred, B, HWh)
        else:
            logits = self._act(plan, "logits", Mh, self.n_bins)
            self._lin("conv_out.w", ram, logits, Mh, hip.ACT_NONE, None, "conv_out")
            ops.bin_softmax(logits, centers, prob, pred, B, HWh, self.n_bins)
        if taps is not None:
            taps["unet_out"] = self._nchw(unet, B, hs[0], wsz[0])
            taps["ram"] = self._nchw(ram, B, hs[0], wsz[0])
        return edges, pred, prob
