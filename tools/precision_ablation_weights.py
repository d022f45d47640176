#!/usr/bin/env python3
"""Weight-rounding error by parameter group: the f32 engine with ONE group of GEMM / conv weights rounded to the 16-bit format,
relative L1 of `pred` against the unrounded f32 engine (B=1, 480x640, bench input)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine

GROUPS = {
    "encoder.stem+stage0-2": lambda k: k.startswith(("img_encoder.conv0", "img_encoder.conv1", "img_encoder.conv2")),
    "encoder.stage3-5 (IR)": lambda k: k.startswith(("img_encoder.conv3", "img_encoder.conv4")),
    "hist_encoder": lambda k: k.startswith("hist_encoder"),
    "decoder.up1-4+conv1-4": lambda k: k.startswith(("decoder.up", "decoder.conv4", "decoder.conv3", "decoder.conv2", "decoder.conv1")),
    "fusion (cross_atten*)": lambda k: k.startswith("decoder.cross_atten"),
    "decoder.conv0": lambda k: k.startswith("decoder.conv0"),
    "depth_head.conv3x3": lambda k: k.startswith("depth_head.conv3x3"),
    "conv_out": lambda k: k.startswith("conv_out"),
}


def rel(a, b):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    return float(np.abs(a - b).sum() / np.abs(a).sum())


def roundable(k, v):
    return (torch.is_tensor(v) and v.is_floating_point() and v.dim() >= 2 and "positional" not in k and ".se." not in k
            and "regressor" not in k and "conv1x1" not in k)


layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
inp = synthetic.to_device(synthetic.make_inputs(1), "cuda:0")
e32 = Engine(sd, layer_names=layers, dtype=torch.float32)
_, p32, _ = e32.forward(inp)
p32 = p32.clone()
for dt, name in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
    tot = 0.0
    for g, sel in GROUPS.items():
        sdw = {k: (v.to(dt).float() if roundable(k, v) and sel(k) else v) for k, v in sd.items()}
        e32.load_state_dict(sdw)
        _, p, _ = e32.forward(inp)
        r = rel(p32, p)
        tot += r * r
        print(f"{name} weights rounded in {g:28s}: {r:.3e}")
    print(f"{name} root-sum-square of the groups: {tot ** 0.5:.3e}")
