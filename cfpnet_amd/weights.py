"""Key-addressed deterministic parameters.

There is no network on either box, so neither the timm-pretrained encoder nor the authors'
checkpoint can be fetched (reference: `src/models/encoder.py:57`, README.md:22).  Tests and the
benchmark therefore use *synthetic* parameters that any machine can regenerate bit-for-bit
from nothing but the state-dict key: value[i] = f(splitmix64(fnv1a(key) + i)).  The golden
generator (`oracle/gen_golden.py`) loads exactly these tensors into the reference modules, so
fixtures only need to store outputs.

The scale of each tensor is chosen per "init kind" (see `cfpnet_amd/spec.py`) so that
activations stay O(1) through the ~150 layers: the network is random but numerically
well-conditioned, which keeps the parity tests meaningful.
"""
from __future__ import annotations

import math
import re
from typing import Dict, Iterable, Tuple

import numpy as np

_M64 = (1 << 64) - 1


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _M64
    return h


def splitmix64_uniform(seed: int, n: int) -> np.ndarray:
    """n doubles in [0,1): element i is splitmix64's output for state seed + (i+1)*golden."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform(key: str, shape, lo: float, hi: float, salt: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = splitmix64_uniform((fnv1a64(key) + salt) & _M64, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


# gain = (target pre-activation variance) / (input second moment); tuned with
# `tools/rms_report.py` so that stage outputs stay O(1).
_GAIN = {
    "conv_act": 2.4,     # feeds BN + SiLU/ReLU
    "conv_lrelu": 2.0,   # feeds BN + LeakyReLU
    "conv_lin": 1.0,     # linear output (projection, 1x1 skip-less)
    "conv_res": 0.25,    # last conv of a residual branch
    "conv_logit": 4.0,   # bin logits: want a peaked-but-not-one-hot softmax
    "lin": 1.0,
    "lin_act": 2.0,
    "lin_lrelu": 2.0,
    "lin_res": 0.25,
}


# per-key gain overrides (first match wins): places where the architecture itself amplifies --
# every hist2image layer doubles the in-zone tokens (fusion.py:157 adds `message + x` onto x) --
# are followed by a deliberately small conv so the random network stays O(1) end to end.
_OVERRIDES = [
    (re.compile(r"^hist_encoder\.hist_extractor1\.pointnet_encoder\.conv1\.weight$"), 0.3),
    (re.compile(r"^hist_encoder\..*\.conv\d\.weight$"), 2.2),
    (re.compile(r"^decoder\.up[234]\._net\.0\.weight$"), 0.12),
    (re.compile(r"^img_encoder\.conv(3\.\d|4)\.\d+\.(conv_pw|conv_dw)\.weight$"), 2.6),
    (re.compile(r"^img_encoder\.conv(3\.\d|4)\.0\.conv_pwl\.weight$"), 0.9),
    (re.compile(r"^img_encoder\..*\.(conv|conv_exp)\.weight$"), 1.7),
    (re.compile(r"^conv_out\.0\.weight$"), 14.0),
]


def gain_for(key: str, kind: str) -> float:
    for rx, g in _OVERRIDES:
        if rx.match(key):
            return g
    return _GAIN[kind]


def make_tensor(key: str, shape: Tuple[int, ...], kind: str) -> np.ndarray:
    if kind in _GAIN:
        fan_in = int(np.prod(shape[1:]))
        a = math.sqrt(3.0 * gain_for(key, kind) / fan_in)
        return uniform(key, shape, -a, a)
    if kind == "bias":
        return uniform(key, shape, -0.05, 0.05)
    if kind == "bn_weight":
        return uniform(key, shape, 0.8, 1.2)
    if kind == "bn_bias":
        return uniform(key, shape, -0.1, 0.1)
    if kind == "bn_mean":
        return uniform(key, shape, -0.1, 0.1)
    if kind == "bn_var":
        return uniform(key, shape, 0.8, 1.2)
    if kind == "bn_count":
        return np.asarray(1, dtype=np.int64)
    if kind == "ln_weight":
        return uniform(key, shape, 0.8, 1.2)
    if kind == "ln_bias":
        return uniform(key, shape, -0.1, 0.1)
    if kind == "posenc":
        return uniform(key, shape, -0.35, 0.35)   # ~ trunc_normal(std=0.2) spread
    raise KeyError(kind)


def make_state_dict(manifest: Iterable[Tuple[str, Tuple[int, ...], str]]) -> Dict[str, "np.ndarray"]:
    return {k: make_tensor(k, s, kind) for k, s, kind in manifest}


def make_torch_state_dict(manifest):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in make_state_dict(manifest).items()}
