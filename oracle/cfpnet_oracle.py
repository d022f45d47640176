"""CPU oracle for the CFPNet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-CPU fp32 restatement of the reference's inference forward
(`/root/reference/src/models/deltar.py:34-67`), written as flat functions over a state-dict.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; `cfpnet_amd/` never does.  The reference itself never travels to the GPU box -- this
file is what the HIP path is compared against there.

Parity status
  * Decoder + fusion + ToF histogram encoder + depth head + bin maths: PINNED against outputs of
    the reference itself (imported in the build container by `oracle/gen_golden.py`, fixtures in
    `tests/golden/`, checked by `tests/test_oracle_golden.py`).
  * RGB encoder (`img_encoder.*`): PARITY UNPINNED.  The arithmetic lives in timm==0.5.4
    (`tf_efficientnetv2_b3`, requirements.txt:51; call site encoder.py:57), which is neither
    vendored in the reference nor installed/installable here.  `encoder()` restates the
    published architecture; what pins it is the parameter manifest
    (shape-for-shape equal to timm's 14.36 M minus head) and the tap shapes decoder.py:67 needs.

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------------
BN_TRAIN = False        # tests flip this to get the model.train() forward (batch statistics, running stats updated in `sd`)


def _bn(sd: SD, p: str, x: Tensor, eps: float = 1e-5, momentum: float = 0.1) -> Tensor:
    """BatchNorm: running statistics (inference) or, with BN_TRAIN, batch statistics like model.train()."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], BN_TRAIN, momentum if BN_TRAIN else 0.0, eps)


def _same_pad(size: int, k: int, s: int) -> Tuple[int, int]:
    """TensorFlow 'SAME' padding (timm Conv2dSame): total = max((ceil(i/s)-1)*s + k - i, 0),
    split floor/ceil between leading and trailing edge."""
    total = max((math.ceil(size / s) - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def _conv_same(x: Tensor, w: Tensor, stride: int, groups: int = 1) -> Tensor:
    k = w.shape[-1]
    pt, pb = _same_pad(x.shape[2], k, stride)
    pl, pr = _same_pad(x.shape[3], k, stride)
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, None, stride, 0, 1, groups)


# ----------------------------------------------------------------------------------------
# E0: RGB encoder  (encoder.py:54-79; architecture: timm 0.5.4 tf_efficientnetv2_b3)
# ----------------------------------------------------------------------------------------
_ENC_STAGES = [("conv0.2", 1), ("conv1", 2), ("conv2", 2), ("conv3.0", 2), ("conv3.1", 1), ("conv4", 2)]
_ENC_EPS = 1e-3
_ENC_MOM = 0.01      # timm tf_ models: bn_momentum = 1 - 0.99 (only matters for the running statistics in BN_TRAIN mode)


def encoder(sd: SD, x: Tensor, stem_act: bool = False) -> List[Tensor]:
    """rgb [B,3,H,W] -> five feature maps (16@1/2, 40@1/4, 56@1/8, 136@1/16, 232@1/32).

    `stem_act=False` reproduces the reference under its pinned timm 0.5.4: encoder.py:58-61
    takes `conv_stem, bn1, blocks[0]` and so skips the separate `act1` SiLU."""
    p = "img_encoder"
    x = _bn(sd, f"{p}.conv0.1", _conv_same(x, sd[f"{p}.conv0.0.weight"], 2), _ENC_EPS, _ENC_MOM)
    if stem_act:
        x = F.silu(x)
    taps = []
    for stage, stride in _ENC_STAGES:
        i = 0
        while any(k.startswith(f"{p}.{stage}.{i}.") for k in sd):
            q = f"{p}.{stage}.{i}"
            s = stride if i == 0 else 1
            inp = x
            if f"{q}.conv.weight" in sd:                       # ConvBnAct
                x = F.silu(_bn(sd, q + ".bn1", _conv_same(x, sd[q + ".conv.weight"], s), _ENC_EPS, _ENC_MOM))
            elif f"{q}.conv_exp.weight" in sd:                 # EdgeResidual
                x = F.silu(_bn(sd, q + ".bn1", _conv_same(x, sd[q + ".conv_exp.weight"], s), _ENC_EPS, _ENC_MOM))
                x = _bn(sd, q + ".bn2", F.conv2d(x, sd[q + ".conv_pwl.weight"]), _ENC_EPS, _ENC_MOM)
            else:                                              # InvertedResidual + SE
                x = F.silu(_bn(sd, q + ".bn1", F.conv2d(x, sd[q + ".conv_pw.weight"]), _ENC_EPS, _ENC_MOM))
                w = sd[q + ".conv_dw.weight"]
                x = F.silu(_bn(sd, q + ".bn2", _conv_same(x, w, s, groups=w.shape[0]), _ENC_EPS, _ENC_MOM))
                g = x.mean((2, 3), keepdim=True)
                g = F.silu(F.conv2d(g, sd[q + ".se.conv_reduce.weight"], sd[q + ".se.conv_reduce.bias"]))
                g = F.conv2d(g, sd[q + ".se.conv_expand.weight"], sd[q + ".se.conv_expand.bias"])
                x = x * torch.sigmoid(g)
                x = _bn(sd, q + ".bn3", F.conv2d(x, sd[q + ".conv_pwl.weight"]), _ENC_EPS, _ENC_MOM)
            if s == 1 and inp.shape[1] == x.shape[1]:
                x = x + inp
            i += 1
        if stage != "conv3.0":          # conv3 = blocks[3] + blocks[4]: one tap after both
            taps.append(x)
    return taps


# ----------------------------------------------------------------------------------------
# H0: ToF histogram encoder  (encoder.py:6-50)
# ----------------------------------------------------------------------------------------
def hist_encoder(sd: SD, hist_data: Tensor) -> List[Tensor]:
    """hist_data [B,Z,N] -> [B,Z,N,32], [B,Z,N,64], [B,Z,N,128]; every sample point goes through
    9 x (pointwise conv + BN + ReLU) independently."""
    B, Z, N = hist_data.shape
    x = hist_data.reshape(B * Z * N, 1)
    outs = []
    for e in (1, 2, 3):
        q = f"hist_encoder.hist_extractor{e}.pointnet_encoder"
        for j in (1, 2, 3):
            w = sd[f"{q}.conv{j}.weight"][:, :, 0]
            x = x @ w.t() + sd[f"{q}.conv{j}.bias"]
            x = F.batch_norm(x, sd[f"{q}.bn{j}.running_mean"], sd[f"{q}.bn{j}.running_var"],
                             sd[f"{q}.bn{j}.weight"], sd[f"{q}.bn{j}.bias"], BN_TRAIN, 0.1 if BN_TRAIN else 0.0, 1e-5)
            x = F.relu(x)
        outs.append(x.reshape(B, Z, N, -1))
    return outs


# ----------------------------------------------------------------------------------------
# A0: linear attention  (attention.py:20-52)
# ----------------------------------------------------------------------------------------
def linear_attention(q: Tensor, k: Tensor, v: Tensor, eps: float = 1e-6) -> Tensor:
    """q [N,L,h,d], k/v [N,S,h,d] -> [N,L,h,d] with feature map elu(x)+1.  The v/S ... *S pair
    is kept because it changes fp32 rounding (attention.py:41-42,49)."""
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    S = v.shape[1]
    v = v / S
    Kh, Vh, Qh = K.permute(0, 2, 3, 1), v.permute(0, 2, 1, 3), Q.permute(0, 2, 1, 3)
    KV = Kh @ Vh                                    # [N,h,d,d]
    Ksum = K.sum(dim=1)                             # [N,h,d]
    Zinv = 1.0 / ((Qh * Ksum[:, :, None, :]).sum(-1) + eps)   # [N,h,L]
    out = (Qh @ KV) * Zinv[..., None] * S           # [N,h,L,d]
    return out.permute(0, 2, 1, 3).contiguous()


# ----------------------------------------------------------------------------------------
# A1: LoFTR encoder layer  (transformer.py:41-71)
# ----------------------------------------------------------------------------------------
def loftr_layer(sd: SD, p: str, x: Tensor, source: Tensor, nhead: int) -> Tensor:
    N, L, D = x.shape
    d = D // nhead
    q = (x @ sd[p + ".q_proj.weight"].t()).view(N, -1, nhead, d)
    k = (source @ sd[p + ".k_proj.weight"].t()).view(N, -1, nhead, d)
    v = (source @ sd[p + ".v_proj.weight"].t()).view(N, -1, nhead, d)
    msg = linear_attention(q, k, v).reshape(N, L, D)
    msg = msg @ sd[p + ".merge.weight"].t()
    msg = F.layer_norm(msg, (D,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    h = torch.cat([x, msg], dim=2) @ sd[p + ".mlp.0.weight"].t()
    h = F.relu(h) @ sd[p + ".mlp.2.weight"].t()
    h = F.layer_norm(h, (D,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    return h + x


# ----------------------------------------------------------------------------------------
# X4 / X5: Twins LSA + GSA  (transformer.py:89-116,138-150,154-165)
# ----------------------------------------------------------------------------------------
TWINS_HEADS = 8   # TwinsTransformer does not forward num_heads (transformer.py:157-158)


def lsa(sd: SD, p: str, tok: Tensor, H: int, W: int, ws: int) -> Tensor:
    """Windowed self-attention.  Features are zero-padded *before* the projections, so padded
    tokens still enter the key sum with K = elu(0)+1 = 1 (transformer.py:101-107)."""
    B, _, C = tok.shape
    pb, pr = (ws - H % ws) % ws, (ws - W % ws) % ws
    x = F.pad(tok.view(B, H, W, C), (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, W + pr
    nh, nw = Hp // ws, Wp // ws
    x = x.view(B, nh, ws, nw, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nh * nw, ws * ws, C)
    x = loftr_layer(sd, p + ".encoder_layer", x, x, TWINS_HEADS)
    x = x.view(B, nh, nw, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    return x[:, :H, :W, :].reshape(B, H * W, C)


def gsa(sd: SD, p: str, tok: Tensor, H: int, W: int, ws: int) -> Tensor:
    """Keys/values from a stride-ws, kernel-ws conv (floors) + LayerNorm; queries are all tokens."""
    B, _, C = tok.shape
    x = tok.transpose(1, 2).reshape(B, C, H, W)
    x = F.conv2d(x, sd[p + ".sr.weight"], sd[p + ".sr.bias"], stride=ws)
    x = x.reshape(B, C, -1).transpose(1, 2)
    x = F.layer_norm(x, (C,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
    return loftr_layer(sd, p + ".encoder_layer", tok, x, TWINS_HEADS)


# ----------------------------------------------------------------------------------------
# X2: DAPM  (transformer.py:204-248)   X3: LKPM  (convnext.py:42-58)
# ----------------------------------------------------------------------------------------
def dapm(sd: SD, p: str, tok: Tensor, H: int, W: int, rect: Tuple[int, int, int, int], nhead: int) -> Tensor:
    """Outside-zone tokens query the inside-zone tokens (linear attention, no merge/MLP/LN --
    those parameters are dead), the message is scattered into a zero map, concatenated with the
    features and sent through conv3x3-BN-conv3x3-BN (no activation), plus the residual."""
    B, _, D = tok.shape
    y0, y1, x0, x1 = rect
    d = D // nhead
    grid = tok.view(B, H, W, D)
    inside = grid[:, y0:y1, x0:x1, :].reshape(B, -1, D)
    q = (tok @ sd[p + ".q_proj.weight"].t()).view(B, -1, nhead, d)
    k = (inside @ sd[p + ".k_proj.weight"].t()).view(B, -1, nhead, d)
    v = (inside @ sd[p + ".v_proj.weight"].t()).view(B, -1, nhead, d)
    msg = linear_attention(q, k, v).reshape(B, H, W, D).clone()
    msg[:, y0:y1, x0:x1, :] = 0          # only outside tokens receive a message
    f = torch.cat([grid, msg], dim=3).permute(0, 3, 1, 2)
    f = _bn(sd, p + ".bn1", F.conv2d(f, sd[p + ".conv1.weight"], None, 1, 1))
    f = _bn(sd, p + ".bn2", F.conv2d(f, sd[p + ".conv2.weight"], None, 1, 1))
    return f.permute(0, 2, 3, 1).reshape(B, H * W, D) + tok


def lkpm(sd: SD, p: str, tok: Tensor, H: int, W: int) -> Tensor:
    """Large-kernel depthwise conv -> BN -> ReLU -> LayerNorm(1e-6) -> Linear 4x -> GELU(erf)
    -> Linear -> + input."""
    B, _, D = tok.shape
    x = tok.view(B, H, W, D).permute(0, 3, 1, 2)
    w = sd[p + ".dwconv2.weight"]
    y = F.conv2d(x, w, sd[p + ".dwconv2.bias"], 1, (w.shape[-1] - 1) // 2, 1, D)
    y = F.relu(_bn(sd, p + ".bn1", y)).permute(0, 2, 3, 1)
    y = F.layer_norm(y, (D,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    y = F.gelu(y @ sd[p + ".pwconv1.weight"].t() + sd[p + ".pwconv1.bias"])
    y = y @ sd[p + ".pwconv2.weight"].t() + sd[p + ".pwconv2.bias"]
    return tok + y.reshape(B, H * W, D)


# ----------------------------------------------------------------------------------------
# G2 + P0 + X1: fusion module  (fusion.py:52-188)
# ----------------------------------------------------------------------------------------
def _batch_geometry(patch_info, key):
    """fusion.py:70-84 on the collated (batched, host-side) patch_info."""
    info = patch_info[key] if key in patch_info else patch_info[int(key)]
    t = lambda a: torch.as_tensor(a)
    zn = int(t(patch_info["zone_num"]).reshape(-1)[0])
    pad = t(info["pad_size"]).reshape(-1, 2)
    ps = t(info["patch_size"]).reshape(-1, 2)
    idx = t(info["index_wo_pad"]).reshape(-1, 4)
    pad_h, pad_w = int(pad[:, 0].max()), int(pad[:, 1].max())
    p1, p2 = int(ps[:, 0].max()), int(ps[:, 1].max())
    sy, sx = int(idx[:, 0].min()), int(idx[:, 1].min())
    ey, ex = int(idx[:, 2].max()), int(idx[:, 3].max())
    return zn, pad_h, pad_w, p1, p2, sy, sx, ey, ex


def fusion(sd: SD, p: str, x: Tensor, feat1: Tensor, mask: Tensor, patch_info, *,
           max_resolution: Tuple[int, int], layer_names: Sequence[str], patch_key: float, change_embedding: bool = True,
           no_skip_inside: bool = False, pos_offset: Tuple[int, int] = (0, 0),
           taps: Optional[dict] = None) -> Tensor:
    """x [B,D,H,W] image features, feat1 [B,Z,N,D] ToF embeddings, mask [B,Z] zone validity.

    `pos_offset` is the (y, x) window into the learned positional table; the reference draws it
    with torch.randint when H < Hmax (fusion.py:87-91) -- callers pass the drawn values."""
    B, D, H, W = x.shape
    Hm, Wm = max_resolution
    ws = math.ceil(math.sqrt(math.sqrt(Hm * Wm)))
    zn, pad_h, pad_w, p1, p2, sy, sx, ey, ex = _batch_geometry(patch_info, patch_key)
    tzh, tzw = ey - sy, ex - sx
    interp = tzh != p1 * zn or tzw != p2 * zn
    cy0, cy1 = min(max(sy, 0), H), min(max(ey, 0), H)
    cx0, cx1 = min(max(sx, 0), W), min(max(ex, 0), W)

    oy, ox = pos_offset
    pe = sd[p + ".positional_encodings"].view(Hm, Wm, D)[oy:oy + H, ox:ox + W, :]
    emb0 = x + pe.permute(2, 0, 1)
    tok = emb0.flatten(2).transpose(1, 2).contiguous()           # [B,HW,D]
    src = (feat1 + sd[p + ".positional_encodings2"]).reshape(B * feat1.shape[1], feat1.shape[2], D)
    valid = mask.reshape(-1).to(tok.dtype)                        # [(B Z)]

    for i, name in enumerate(layer_names):
        q = f"{p}.layers.{i}"
        if name == "image":
            tok = lsa(sd, q + ".lga", tok, H, W, ws)
            tok = gsa(sd, q + ".gsa", tok, H, W, ws)
        elif name == "hist2image":
            grid = tok.transpose(1, 2).reshape(B, D, H, W) if change_embedding else emb0
            # rectangle [sy:ey, sx:ex] of the zero-extended map (fusion.py:136-138)
            z = F.pad(grid, (pad_w, pad_w, pad_h, pad_h))[:, :, sy + pad_h:ey + pad_h, sx + pad_w:ex + pad_w]
            if interp:
                z = F.interpolate(z, size=[zn * p1, zn * p2], mode="bilinear", align_corners=True)
            z = z.reshape(B, D, zn, p1, zn, p2).permute(0, 2, 4, 3, 5, 1).reshape(B * zn * zn, p1 * p2, D)
            z = loftr_layer(sd, q, z, src, 4)
            z = z * valid[:, None, None]                            # zero zones without ToF signal
            z = z.reshape(B, zn, zn, p1, p2, D).permute(0, 5, 1, 3, 2, 4).reshape(B, D, zn * p1, zn * p2)
            if interp:
                z = F.interpolate(z, size=[tzh, tzw], mode="bilinear", align_corners=True)
            z = z[:, :, cy0 - sy:cy1 - sy, cx0 - sx:cx1 - sx].permute(0, 2, 3, 1)   # part inside the image
            g = tok.view(B, H, W, D).clone()
            if no_skip_inside:
                g[:, cy0:cy1, cx0:cx1, :] = z
            else:
                g[:, cy0:cy1, cx0:cx1, :] += z
            tok = g.view(B, H * W, D)
        elif name == "combine1":
            tok = dapm(sd, q + ".transformer_path", tok, H, W, (cy0, cy1, cx0, cx1), 4)
            tok = lkpm(sd, q + ".large_kernel_path", tok, H, W)
        else:
            raise NotImplementedError(name)
        if taps is not None:
            taps[f"{p}.layers.{i}"] = tok
    return tok.transpose(1, 2).reshape(B, D, H, W).contiguous()


# ----------------------------------------------------------------------------------------
# U0 / U1: decoder  (decoder.py:51-58,96-128)
# ----------------------------------------------------------------------------------------
_FUSION_STRIDE = {"cross_atten1": 4, "cross_atten2": 8, "cross_atten3": 16}      # max_resolution = base / stride (decoder.py:82-94)


def _up(sd: SD, p: str, x: Tensor, skip: Tensor) -> Tensor:
    x = F.interpolate(x, size=[skip.shape[2], skip.shape[3]], mode="bilinear", align_corners=True)
    x = torch.cat([x, skip], dim=1)
    for c, b in ((0, 1), (3, 4)):
        x = F.conv2d(x, sd[f"{p}._net.{c}.weight"], sd[f"{p}._net.{c}.bias"], 1, 1)
        x = F.leaky_relu(_bn(sd, f"{p}._net.{b}", x), 0.01)
    return x


def decoder(sd: SD, img_features: Sequence[Tensor], hist_features: Sequence[Tensor], mask: Tensor,
            patch_info, *, layer_names, change_embedding=True, no_skip_inside=False,
            pos_offsets: Optional[dict] = None, taps: Optional[dict] = None, base_resolution=(480, 640)) -> Tensor:
    b0, b1, b2, b3, b4 = img_features
    f1, f2, f3 = hist_features
    pos_offsets = pos_offsets or {}
    kw = dict(layer_names=layer_names, change_embedding=change_embedding, no_skip_inside=no_skip_inside, taps=taps)

    def fuse(name, x, feat):
        s = _FUSION_STRIDE[name]
        Hm, Wm = base_resolution[0] // s, base_resolution[1] // s
        # fusion.py:41: patch_info is indexed with 640 / max_resolution[1], i.e. the stride of the scale
        return fusion(sd, f"decoder.{name}", x, feat, mask, patch_info, max_resolution=(Hm, Wm), patch_key=base_resolution[1] / Wm,
                      pos_offset=pos_offsets.get(name, (0, 0)), **kw)

    def rec(k, v):
        if taps is not None:
            taps[k] = v
        return v

    x = F.conv2d(b4, sd["decoder.conv4.weight"], sd["decoder.conv4.bias"])
    x = rec("up1", _up(sd, "decoder.up1", x, b3))
    x = rec("conv3", F.conv2d(x, sd["decoder.conv3.weight"], sd["decoder.conv3.bias"]))
    x = torch.cat([x, rec("cross_atten3", fuse("cross_atten3", x, f3))], dim=1)
    x = rec("up2", _up(sd, "decoder.up2", x, b2))
    x = rec("conv2", F.conv2d(x, sd["decoder.conv2.weight"], sd["decoder.conv2.bias"]))
    x = torch.cat([x, rec("cross_atten2", fuse("cross_atten2", x, f2))], dim=1)
    x = rec("up3", _up(sd, "decoder.up3", x, b1))
    x = rec("conv1", F.conv2d(x, sd["decoder.conv1.weight"], sd["decoder.conv1.bias"]))
    x = torch.cat([x, rec("cross_atten1", fuse("cross_atten1", x, f1))], dim=1)
    x = rec("up4", _up(sd, "decoder.up4", x, b0))
    return rec("unet_out", F.conv2d(x, sd["decoder.conv0.weight"], sd["decoder.conv0.bias"], 1, 1))


# ----------------------------------------------------------------------------------------
# R0 / R1: adaptive-bin head  (decoder.py:22-37, deltar.py:50-61)
# ----------------------------------------------------------------------------------------
def depth_head(sd: SD, x: Tensor, norm: str = "linear") -> Tuple[Tensor, Tensor]:
    ram = F.conv2d(x, sd["depth_head.conv3x3.weight"], sd["depth_head.conv3x3.bias"], 1, 1)
    y = F.conv2d(x, sd["depth_head.conv1x1.weight"]).mean([2, 3])
    y = F.leaky_relu(y @ sd["depth_head.regressor.0.weight"].t() + sd["depth_head.regressor.0.bias"], 0.01)
    y = F.leaky_relu(y @ sd["depth_head.regressor.2.weight"].t() + sd["depth_head.regressor.2.bias"], 0.01)
    y = y @ sd["depth_head.regressor.4.weight"].t() + sd["depth_head.regressor.4.bias"]
    if norm == "linear":
        y = torch.relu(y) + 0.1
    elif norm == "softmax":
        return torch.softmax(y, dim=1), ram
    else:
        y = torch.sigmoid(y)
    return y / y.sum(dim=1, keepdim=True), ram


def bins_to_depth(sd: SD, widths_normed: Tensor, ram: Tensor, min_val: float, max_val: float):
    prob = torch.softmax(F.conv2d(ram, sd["conv_out.0.weight"], sd["conv_out.0.bias"]), dim=1)
    widths = F.pad((max_val - min_val) * widths_normed, (1, 0), mode="constant", value=min_val)
    edges = torch.cumsum(widths, dim=1)
    centers = 0.5 * (edges[:, :-1] + edges[:, 1:])
    pred = (prob * centers[:, :, None, None]).sum(dim=1, keepdim=True)
    return edges, pred, prob


# ----------------------------------------------------------------------------------------
# whole model
# ----------------------------------------------------------------------------------------
def forward(sd: SD, input_data: dict, *, layer_names, min_val: float = 1e-3, max_val: float = 10.0,
            norm: str = "linear", change_embedding: bool = True, no_skip_inside: bool = False,
            stem_act: bool = False, pos_offsets: Optional[dict] = None, taps: Optional[dict] = None,
            img_features: Optional[Sequence[Tensor]] = None, grad: bool = False, base_resolution=(480, 640)):
    """Eval-mode `Deltar.forward` (deltar.py:34-67): returns (bin_edges, pred, prob).
    `img_features` bypasses the RGB encoder (used to pin everything else against the reference)."""
    add = input_data["additional"]
    with torch.set_grad_enabled(grad):     # grad=True (+ BN_TRAIN): the training forward, differentiated by autograd in the tests
        feats = list(img_features) if img_features is not None else encoder(sd, input_data["rgb"], stem_act)
        if taps is not None:
            for i, f in enumerate(feats):
                taps[f"enc{i}"] = f
        hfeat = hist_encoder(sd, add["hist_data"])
        if taps is not None:
            for i, f in enumerate(hfeat):
                taps[f"hist{i}"] = f
        unet = decoder(sd, feats, hfeat, add["mask"], add["patch_info"], layer_names=layer_names,
                       change_embedding=change_embedding, no_skip_inside=no_skip_inside,
                       pos_offsets=pos_offsets, taps=taps, base_resolution=base_resolution)
        widths, ram = depth_head(sd, unet, norm)
        if taps is not None:
            taps["ram"] = ram
            taps["widths"] = widths
        return bins_to_depth(sd, widths, ram, min_val, max_val)


# ----------------------------------------------------------------------------------------
# L0: SILog loss (loss.py:9-19) and the evaluation metrics (utils/metrics.py:4-24)
# ----------------------------------------------------------------------------------------
def silog_loss(pred: Tensor, target: Tensor, mask: Optional[Tensor] = None, interpolate: bool = True) -> Tensor:
    if interpolate:
        pred = F.interpolate(pred, target.shape[-2:], mode="bilinear", align_corners=True)
    if mask is not None:
        pred, target = pred[mask], target[mask]
    g = torch.log(pred) - torch.log(target)
    return 10 * torch.sqrt(torch.var(g) + 0.15 * torch.mean(g) ** 2)


def compute_errors(gt, pred) -> dict:
    import numpy as np
    thresh = np.maximum(gt / pred, pred / gt)
    err = np.log(pred) - np.log(gt)
    return dict(
        a1=(thresh < 1.25).mean(), a2=(thresh < 1.25 ** 2).mean(), a3=(thresh < 1.25 ** 3).mean(),
        abs_rel=np.mean(np.abs(gt - pred) / gt), rmse=np.sqrt(((gt - pred) ** 2).mean()),
        log_10=np.abs(np.log10(gt) - np.log10(pred)).mean(),
        rmse_log=np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean()),
        silog=np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100,
        sq_rel=np.mean(((gt - pred) ** 2) / gt))
