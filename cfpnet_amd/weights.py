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
from typing import Dict, Iterable, Optional, Tuple

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


def normal(key: str, shape, std: float, salt: int = 0) -> np.ndarray:
    """Key-addressed N(0, std^2): Box-Muller on two splitmix64 streams of the key."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = splitmix64_uniform((fnv1a64(key) + 2 * salt + 1) & _M64, n)
    u2 = splitmix64_uniform((fnv1a64(key) + 2 * salt + 2) & _M64, n)
    z = np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * math.pi * u2)
    return (std * z).astype(np.float32).reshape(shape)


def make_tensor_kaiming(key: str, shape: Tuple[int, ...], kind: str) -> np.ndarray:
    """The reference's OWN initialisation family (`Deltar._reset_parameters`, deltar.py:23-32; fusion.py:22-23): every Conv / Linear
    outside the RGB encoder is kaiming-normal with fan_out / relu gain (std = sqrt(2 / (Cout * kh * kw))), biases keep PyTorch's default
    U(+-1/sqrt(fan_in)), BatchNorm / LayerNorm affine parameters are (1, 0), the positional tables trunc_normal(std 0.2).  The encoder
    is not re-initialised by the reference (timm's pretrained weights, unavailable offline): it keeps the key-addressed uniform
    tensors.  BatchNorm RUNNING statistics are placeholders here; the tests calibrate them to the network's own batch statistics
    (what training leaves in them), tests/helpers.calibrate_bn."""
    if key.startswith("img_encoder."):
        return make_tensor(key, shape, kind)
    if kind in _GAIN:
        fan_out = int(shape[0] * np.prod(shape[2:])) if len(shape) > 2 else int(shape[0])
        return normal(key, shape, math.sqrt(2.0 / fan_out))
    if kind == "bias":
        return uniform(key, shape, -0.05, 0.05)
    if kind in ("bn_weight", "ln_weight"):
        return np.ones(shape, dtype=np.float32)
    if kind in ("bn_bias", "ln_bias"):
        return np.zeros(shape, dtype=np.float32)
    if kind == "posenc":
        return np.clip(normal(key, shape, 0.2), -2.0, 2.0)
    return make_tensor(key, shape, kind)


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


def make_tensor_kaiming_peaked(key: str, shape: Tuple[int, ...], kind: str) -> np.ndarray:
    """The kaiming family with a CONFIDENT head: conv_out's weights x 6, so the 256-way softmax is peaked like a trained model's
    (the plain family's is nearly flat -- max prob 0.03 -- which hides logit errors in the expectation)."""
    t = make_tensor_kaiming(key, shape, kind)
    return t * np.float32(6.0) if key == "conv_out.0.weight" else t


FAMILIES = {"uniform": make_tensor, "kaiming": make_tensor_kaiming, "kaiming_peaked": make_tensor_kaiming_peaked}


def make_state_dict(manifest: Iterable[Tuple[str, Tuple[int, ...], str]], family: str = "uniform") -> Dict[str, "np.ndarray"]:
    f = FAMILIES[family]
    return {k: f(k, s, kind) for k, s, kind in manifest}


def make_torch_state_dict(manifest, family: str = "uniform"):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in make_state_dict(manifest, family).items()}


def trained_like_state_dict(layer_names, steps: int = 300, batch: int = 4, device="cuda:0", lr: float = 3e-4, seed: int = 7, dtype=None,
                            loss_log: Optional[list] = None):
    """The closest stand-in for the authors' `best.pt` (/root/reference/README.md:22) that can be made offline: the reference's own
    initialisation (the `kaiming` family) after `steps` real optimisation steps of `trainer.Trainer` (train.py:104-135: training forward,
    SILog, backward, AdamW / OneCycle) on synthetic 416x544 crops whose target depth FOLLOWS the ToF zones (zone means of the histogram
    samples, bilinearly spread over the crop) -- so the network learns depth completion from its inputs: a confident (peaked) 256-way head,
    BatchNorm running statistics that are real moving averages, weights that have left their initial distribution.  Runs on the GPU
    (there is no CPU training path); returns a reference-layout state dict on the CPU.  `dtype`: the Trainer's storage mode (default float32);
    `loss_log`: receives the SILog value of every step."""
    import torch
    import torch.nn.functional as F
    from . import spec, synthetic
    from .trainer import Trainer
    sd = make_torch_state_dict(spec.model_manifest(layer_names), family="kaiming")
    H, W, zn, zpx = 416, 544, 6, 64
    data = []
    for i in range(6):
        inp = synthetic.make_inputs(batch, H, W, zn, zpx, seed=seed + 31 * i, drop_hist=0.1 * (i % 3))
        hist = inp["additional"]["hist_data"].float()                       # [B, Z, 16] sample points of each zone's depth distribution
        zmean = hist.mean(-1).reshape(batch, 1, zn, zn)
        zmean = torch.where(inp["additional"]["mask"].reshape(batch, 1, zn, zn), zmean, zmean.mean((2, 3), keepdim=True))
        tgt = F.interpolate(zmean, size=(H, W), mode="bilinear", align_corners=False).clamp(0.1, 9.5)
        data.append((synthetic.to_device(inp, device), tgt.to(device)))
    tr = Trainer(sd, list(layer_names), lr=lr, total_steps=steps, dtype=dtype or torch.float32, device=device)
    tr.capture(*data[0])
    with torch.random.fork_rng(devices=[]):          # the positional-encoding windows are drawn from torch's CPU generator: same family every run
        torch.manual_seed(seed)
        for i in range(steps):
            loss, _, _ = tr.step(*data[i % len(data)])
            if loss_log is not None:                 # (one host sync per step: the convergence A/B of tests/test_train_step_gpu.py)
                loss_log.append(float(loss))
    torch.cuda.synchronize()
    out = tr.state_dict()
    out["__loss__"] = float(loss)
    del tr
    torch.cuda.empty_cache()
    return out
