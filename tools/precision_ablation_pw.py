#!/usr/bin/env python3
"""Experiment: bracketed error-diffusion rounding (ops.round_taps) along the INPUT-CHANNEL axis of pointwise (1x1 / Linear) weights,
by group, vs round-to-nearest.  f32 engine, only that group's weights rounded, rel-L1 of pred vs the exact f32 engine, two inputs."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import ops, spec, synthetic, weights
from cfpnet_amd.engine import Engine


def rel(a, b):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    return float(np.abs(a - b).sum() / np.abs(a).sum())


def diff_cin(w, dt):
    shp = w.shape
    flat = w.float().reshape(shp[0], 1, -1)          # [Co, 1, Ci]: "taps" = input channels
    return ops.round_taps(flat, dt).reshape(shp)


G = {"encoder IR pw (expand)": lambda k: k.startswith(("img_encoder.conv3", "img_encoder.conv4")) and "conv_pw." in k,
     "encoder IR pwl (project)": lambda k: k.startswith(("img_encoder.conv3", "img_encoder.conv4")) and "conv_pwl." in k,
     "encoder ER pwl": lambda k: k.startswith(("img_encoder.conv1", "img_encoder.conv2")) and "conv_pwl" in k,
     "decoder conv1-4": lambda k: k.startswith(("decoder.conv4", "decoder.conv3", "decoder.conv2", "decoder.conv1")),
     "fusion q/k/v proj": lambda k: k.startswith("decoder.cross_atten") and "_proj" in k,
     "fusion merge": lambda k: k.startswith("decoder.cross_atten") and ".merge." in k and "transformer_path" not in k,
     "fusion mlp.0": lambda k: k.startswith("decoder.cross_atten") and ".mlp.0." in k and "transformer_path" not in k,
     "fusion mlp.2": lambda k: k.startswith("decoder.cross_atten") and ".mlp.2." in k and "transformer_path" not in k,
     "LKPM pwconv1": lambda k: "pwconv1" in k, "LKPM pwconv2": lambda k: "pwconv2" in k,
     "hist_encoder": lambda k: k.startswith("hist_encoder"), "conv_out": lambda k: k.startswith("conv_out"),
     "all pointwise": lambda k: True}
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
e32 = Engine(sd, layer_names=layers, dtype=torch.float32)
pwt = lambda k, v: (torch.is_tensor(v) and v.is_floating_point() and (v.dim() == 2 or (v.dim() in (3, 4) and v.shape[-1] == 1 and v.shape[1] > 1))
                    and "positional" not in k and ".se." not in k and "regressor" not in k and "conv1x1" not in k)
for seed in (synthetic.SEED, 7):
    inp = synthetic.to_device(synthetic.make_inputs(1, seed=seed), "cuda:0")
    e32.load_state_dict(sd)
    p32 = e32.forward(inp)[1].clone()
    for dt, name in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        for g, sel in G.items():
            res = []
            for fn in (lambda w: w.to(dt).float(), lambda w: diff_cin(w, dt)):
                e32.load_state_dict({k: (fn(v) if pwt(k, v) and sel(k) else v) for k, v in sd.items()})
                res.append(rel(p32, e32.forward(inp)[1]))
            print(f"seed {seed} {name} {g:26s}: nearest {res[0]:.3e}  diffused along Cin {res[1]:.3e}  ratio {res[1] / res[0]:.2f}", flush=True)
