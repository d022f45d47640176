"""Architecture tables and the state-dict manifest of the CFPNet hot path.

Everything here is *data*: which layers exist, their shapes and their key names in the
checkpoint format the reference saves (`/root/reference/src/utils/model_io.py:5-17`,
SURVEY.md App. C).  Both the product engine (`cfpnet_amd/engine.py`) and the parameter
container (`cfpnet_amd/deltar.py`) are driven from these tables.

Reference anchors
  * RGB encoder: `src/models/encoder.py:54-79` picks `conv_stem, bn1, blocks[0..5]` out of
    timm 0.5.4 `tf_efficientnetv2_b3` (requirements.txt:51).  timm is not vendored in the
    reference; the block table below restates the published architecture.
  * ToF histogram encoder: `src/models/encoder.py:6-50`.
  * Decoder / fusion: `src/models/decoder.py:59-94`, `src/models/fusion.py:12-41`,
    `src/models/transformer.py:14-39,76-87,119-136,154-158,169-202,252-261`,
    `src/models/convnext.py:28-40`.
  * Head: `src/models/decoder.py:9-20`, `src/models/deltar.py:16-19`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

# --------------------------------------------------------------------------------------
# RGB encoder (tf_efficientnetv2_b3 feature extractor)
# --------------------------------------------------------------------------------------


@dataclass(frozen=True)
class EncBlock:
    """One encoder block.  kind: 'cn' ConvBnAct, 'er' EdgeResidual, 'ir' InvertedResidual+SE."""

    prefix: str   # state-dict prefix below `img_encoder.`
    kind: str
    cin: int
    cout: int
    mid: int      # expanded width (== cout for 'cn')
    stride: int
    se_rd: int    # squeeze width of the SE block (0: no SE)
    skip: bool    # residual connection


def _stage(prefix: str, kind: str, cin: int, cout: int, repeats: int, stride: int, expand: int,
           se: bool) -> List[EncBlock]:
    out = []
    for i in range(repeats):
        ci = cin if i == 0 else cout
        s = stride if i == 0 else 1
        mid = cout if kind == "cn" else ci * expand
        rd = int(round(mid * (0.25 / expand))) if se else 0
        out.append(EncBlock(f"{prefix}.{i}", kind, ci, cout, mid, s, rd, s == 1 and ci == cout))
    return out


ENC_STEM_OUT = 40
ENC_BN_EPS = 1e-3  # tf_ models use the TensorFlow default
# encoder.py:58-69 groups the timm blocks as conv0=[stem,bn1,blocks[0]], conv1=blocks[1],
# conv2=blocks[2], conv3=[blocks[3],blocks[4]], conv4=blocks[5]
ENC_BLOCKS: List[EncBlock] = (
    _stage("conv0.2", "cn", 40, 16, 2, 1, 1, False)
    + _stage("conv1", "er", 16, 40, 3, 2, 4, False)
    + _stage("conv2", "er", 40, 56, 3, 2, 4, False)
    + _stage("conv3.0", "ir", 56, 112, 5, 2, 4, True)
    + _stage("conv3.1", "ir", 112, 136, 7, 1, 6, True)
    + _stage("conv4", "ir", 136, 232, 12, 2, 6, True)
)
# after which block index each of the five feature taps is taken (encoder.py:71-79)
ENC_TAPS = {1: 0, 4: 1, 7: 2, 19: 3, 31: 4}   # block index -> tap number
ENC_TAP_CHANNELS = [16, 40, 56, 136, 232]


def _bn_keys(prefix: str, c: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    return [
        (f"{prefix}.weight", (c,), "bn_weight"),
        (f"{prefix}.bias", (c,), "bn_bias"),
        (f"{prefix}.running_mean", (c,), "bn_mean"),
        (f"{prefix}.running_var", (c,), "bn_var"),
        (f"{prefix}.num_batches_tracked", (), "bn_count"),
    ]


def encoder_manifest() -> List[Tuple[str, Tuple[int, ...], str]]:
    m: List[Tuple[str, Tuple[int, ...], str]] = []
    p = "img_encoder"
    m.append((f"{p}.conv0.0.weight", (ENC_STEM_OUT, 3, 3, 3), "conv_lin"))
    m += _bn_keys(f"{p}.conv0.1", ENC_STEM_OUT)
    for b in ENC_BLOCKS:
        q = f"{p}.{b.prefix}"
        if b.kind == "cn":
            m.append((f"{q}.conv.weight", (b.cout, b.cin, 3, 3), "conv_act"))
            m += _bn_keys(f"{q}.bn1", b.cout)
        elif b.kind == "er":
            m.append((f"{q}.conv_exp.weight", (b.mid, b.cin, 3, 3), "conv_act"))
            m += _bn_keys(f"{q}.bn1", b.mid)
            m.append((f"{q}.conv_pwl.weight", (b.cout, b.mid, 1, 1), "conv_res" if b.skip else "conv_lin"))
            m += _bn_keys(f"{q}.bn2", b.cout)
        else:
            m.append((f"{q}.conv_pw.weight", (b.mid, b.cin, 1, 1), "conv_act"))
            m += _bn_keys(f"{q}.bn1", b.mid)
            m.append((f"{q}.conv_dw.weight", (b.mid, 1, 3, 3), "conv_act"))
            m += _bn_keys(f"{q}.bn2", b.mid)
            m.append((f"{q}.se.conv_reduce.weight", (b.se_rd, b.mid, 1, 1), "conv_act"))
            m.append((f"{q}.se.conv_reduce.bias", (b.se_rd,), "bias"))
            m.append((f"{q}.se.conv_expand.weight", (b.mid, b.se_rd, 1, 1), "conv_lin"))
            m.append((f"{q}.se.conv_expand.bias", (b.mid,), "bias"))
            m.append((f"{q}.conv_pwl.weight", (b.cout, b.mid, 1, 1), "conv_res" if b.skip else "conv_lin"))
            m += _bn_keys(f"{q}.bn3", b.cout)
    return m


# --------------------------------------------------------------------------------------
# ToF histogram encoder
# --------------------------------------------------------------------------------------
HIST_CHANNELS = [32, 64, 128]


def hist_manifest() -> List[Tuple[str, Tuple[int, ...], str]]:
    m = []
    cin = 1
    for e, c in enumerate(HIST_CHANNELS, start=1):
        q = f"hist_encoder.hist_extractor{e}.pointnet_encoder"
        ci = cin
        for j in (1, 2, 3):
            m.append((f"{q}.conv{j}.weight", (c, ci, 1), "conv_act"))
            m.append((f"{q}.conv{j}.bias", (c,), "bias"))
            ci = c
        for j in (1, 2, 3):
            m += _bn_keys(f"{q}.bn{j}", c)
        cin = c
    return m


# --------------------------------------------------------------------------------------
# Decoder + fusion
# --------------------------------------------------------------------------------------
DEC_ENC_CH = [232, 136, 56, 40, 16]      # decoder.py:67
DEC_CH = [256, 256, 128, 64, 32]         # decoder.py:68
# fusion module name -> (embedding dim, max_resolution, large kernel)   decoder.py:82-94
BASE_RESOLUTION = (480, 640)             # decoder.py:82-88: the tables are sized for this image size


def fusion_table(base_resolution=BASE_RESOLUTION):
    """name -> (embedding dim, max_resolution, large kernel) for a model whose positional tables cover images up to
    `base_resolution` (H, W).  The reference hard-codes 480x640 (decoder.py:82-88); BASELINE configs[4] (640x960) is the
    same construction with a larger base -- table, window size and sr-conv kernel all follow from it."""
    bh, bw = base_resolution
    assert bh % 16 == 0 and bw % 16 == 0, base_resolution
    return {
        "cross_atten3": (128, (bh // 16, bw // 16), 7),
        "cross_atten2": (64, (bh // 8, bw // 8), 15),
        "cross_atten1": (32, (bh // 4, bw // 4), 31),
    }


FUSION = fusion_table()
X2I_HEADS = 4     # fusion.py:13,26,35
TWINS_HEADS = 8   # transformer.py:78,122 defaults (TwinsTransformer drops its num_heads arg)


def window_size(max_resolution) -> int:
    """fusion.py:28"""
    return math.ceil(math.sqrt(math.sqrt(max_resolution[0] * max_resolution[1])))


def _loftr_keys(q: str, d: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    return [
        (f"{q}.q_proj.weight", (d, d), "lin"),
        (f"{q}.k_proj.weight", (d, d), "lin"),
        (f"{q}.v_proj.weight", (d, d), "lin"),
        (f"{q}.merge.weight", (d, d), "lin"),
        (f"{q}.mlp.0.weight", (2 * d, 2 * d), "lin_act"),
        (f"{q}.mlp.2.weight", (d, 2 * d), "lin"),
        (f"{q}.norm1.weight", (d,), "ln_weight"),
        (f"{q}.norm1.bias", (d,), "ln_bias"),
        (f"{q}.norm2.weight", (d,), "ln_weight"),
        (f"{q}.norm2.bias", (d,), "ln_bias"),
    ]


def fusion_manifest(name: str, layer_names: List[str], zone_sample_num: int = 16, base_resolution=BASE_RESOLUTION):
    d, maxres, lk = fusion_table(base_resolution)[name]
    q = f"decoder.{name}"
    ws = window_size(maxres)
    m = [
        (f"{q}.positional_encodings", (maxres[0] * maxres[1], d), "posenc"),
        (f"{q}.positional_encodings2", (zone_sample_num, d), "posenc"),
    ]
    for i, ln in enumerate(layer_names):
        l = f"{q}.layers.{i}"
        if ln == "hist2image":
            m += _loftr_keys(l, d)
        elif ln == "image":
            m += _loftr_keys(f"{l}.lga.encoder_layer", d)
            m += _loftr_keys(f"{l}.gsa.encoder_layer", d)
            m.append((f"{l}.gsa.sr.weight", (d, d, ws, ws), "conv_lin"))
            m.append((f"{l}.gsa.sr.bias", (d,), "bias"))
            m.append((f"{l}.gsa.norm.weight", (d,), "ln_weight"))
            m.append((f"{l}.gsa.norm.bias", (d,), "ln_bias"))
        elif ln == "combine1":
            t = f"{l}.transformer_path"
            m += _loftr_keys(t, d)  # merge/mlp/norm* are dead parameters (never used in forward)
            m.append((f"{t}.conv1.weight", (d, 2 * d, 3, 3), "conv_lin"))
            m += _bn_keys(f"{t}.bn1", d)
            m.append((f"{t}.conv2.weight", (d, d, 3, 3), "conv_res"))
            m += _bn_keys(f"{t}.bn2", d)
            k = f"{l}.large_kernel_path"
            m.append((f"{k}.dwconv2.weight", (d, 1, lk, lk), "conv_act"))
            m.append((f"{k}.dwconv2.bias", (d,), "bias"))
            m.append((f"{k}.norm.weight", (d,), "ln_weight"))
            m.append((f"{k}.norm.bias", (d,), "ln_bias"))
            m.append((f"{k}.pwconv1.weight", (4 * d, d), "lin_act"))
            m.append((f"{k}.pwconv1.bias", (4 * d,), "bias"))
            m.append((f"{k}.pwconv2.weight", (d, 4 * d), "lin_res"))
            m.append((f"{k}.pwconv2.bias", (d,), "bias"))
            m.append((f"{k}.conv1.weight", (d, 2 * d, 1, 1), "conv_lin"))  # dead
            m += _bn_keys(f"{k}.bn1", d)
        else:
            raise NotImplementedError(ln)
    return m


DEAD_PARAM_MARKERS = (
    ".transformer_path.merge.", ".transformer_path.mlp.", ".transformer_path.norm1.",
    ".transformer_path.norm2.", ".large_kernel_path.conv1.",
)


def is_dead_param(key: str) -> bool:
    """Parameters that exist in the checkpoint but are never read by forward
    (transformer.py:183-194 vs 204-248; convnext.py:38 vs 42-58)."""
    return any(s in key for s in DEAD_PARAM_MARKERS)


def decoder_manifest(layer_names: List[str], zone_sample_num: int = 16, base_resolution=BASE_RESOLUTION):
    m = []
    e, c = DEC_ENC_CH, DEC_CH
    m.append(("decoder.conv4.weight", (c[0], e[0], 1, 1), "conv_lin"))
    m.append(("decoder.conv4.bias", (c[0],), "bias"))
    for i in range(1, 5):
        cin, co = c[i - 1] + e[i], c[i]
        q = f"decoder.up{i}._net"
        m.append((f"{q}.0.weight", (co, cin, 3, 3), "conv_lrelu"))
        m.append((f"{q}.0.bias", (co,), "bias"))
        m += _bn_keys(f"{q}.1", co)
        m.append((f"{q}.3.weight", (co, co, 3, 3), "conv_lrelu"))
        m.append((f"{q}.3.bias", (co,), "bias"))
        m += _bn_keys(f"{q}.4", co)
    m.append(("decoder.conv3.weight", (c[2], c[1], 1, 1), "conv_lin"))
    m.append(("decoder.conv3.bias", (c[2],), "bias"))
    m.append(("decoder.conv2.weight", (c[3], c[2], 1, 1), "conv_lin"))
    m.append(("decoder.conv2.bias", (c[3],), "bias"))
    m.append(("decoder.conv1.weight", (c[4], c[3], 1, 1), "conv_lin"))
    m.append(("decoder.conv1.bias", (c[4],), "bias"))
    m.append(("decoder.conv0.weight", (128, c[4], 3, 3), "conv_lin"))
    m.append(("decoder.conv0.bias", (128,), "bias"))
    for name in ("cross_atten1", "cross_atten2", "cross_atten3"):
        m += fusion_manifest(name, layer_names, zone_sample_num, base_resolution)
    return m


def head_manifest(n_bins: int = 256):
    return [
        ("depth_head.conv3x3.weight", (128, 128, 3, 3), "conv_lin"),
        ("depth_head.conv3x3.bias", (128,), "bias"),
        ("depth_head.conv1x1.weight", (128, 128, 1, 1), "conv_lin"),
        ("depth_head.regressor.0.weight", (256, 128), "lin_lrelu"),
        ("depth_head.regressor.0.bias", (256,), "bias"),
        ("depth_head.regressor.2.weight", (256, 256), "lin_lrelu"),
        ("depth_head.regressor.2.bias", (256,), "bias"),
        ("depth_head.regressor.4.weight", (n_bins, 256), "lin"),
        ("depth_head.regressor.4.bias", (n_bins,), "bias"),
        ("conv_out.0.weight", (n_bins, 128, 1, 1), "conv_logit"),
        ("conv_out.0.bias", (n_bins,), "bias"),
    ]


COMBINE1_LAYERS = ["hist2image", "combine1", "image", "hist2image", "combine1", "image"]
BASELINE_LAYERS = ["hist2image", "image", "hist2image", "image"]


def model_manifest(layer_names=None, n_bins: int = 256, zone_sample_num: int = 16, base_resolution=BASE_RESOLUTION):
    """(key, shape, init-kind) for every entry of `Deltar.state_dict()` in registration order of
    the reference (deltar.py:14-19): img_encoder, hist_encoder, depth_head, decoder, conv_out."""
    layer_names = list(layer_names or COMBINE1_LAYERS)
    head = head_manifest(n_bins)
    return (encoder_manifest() + hist_manifest() + head[:9]
            + decoder_manifest(layer_names, zone_sample_num, base_resolution) + head[9:])


def param_count(manifest) -> int:
    n = 0
    for _, shape, kind in manifest:
        if kind in ("bn_mean", "bn_var", "bn_count"):
            continue
        k = 1
        for s in shape:
            k *= s
        n += k
    return n
