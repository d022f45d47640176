#!/usr/bin/env python3
"""Weight-rounding error of all k x k convolutions (dense 3x3, depthwise 3x3, large-kernel depthwise, sr convs) under three
roundings: nearest, error diffusion over the taps constrained to the two bracketing values (ops.round_taps, what the engine
packs), unconstrained error diffusion.  f32 engine with only those weights rounded; rel-L1 of pred vs the exact f32 engine."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import ops, spec, synthetic, weights
from cfpnet_amd.engine import Engine


def rel(a, b):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    return float(np.abs(a - b).sum() / np.abs(a).sum())


layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
e32 = Engine(sd, layer_names=layers, dtype=torch.float32)
inp = synthetic.to_device(synthetic.make_inputs(1), "cuda:0")
p32 = e32.forward(inp)[1].clone()
kxk = lambda k, v: torch.is_tensor(v) and v.is_floating_point() and v.dim() == 4 and v.shape[-1] > 1
for dt, name in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
    res = {}
    for mode, fn in (("nearest", lambda w: w.to(dt).float()), ("diffused, bracketed", lambda w: ops.round_taps(w, dt, True)),
                     ("diffused, free", lambda w: ops.round_taps(w, dt, False))):
        e32.load_state_dict({k: (fn(v) if kxk(k, v) else v) for k, v in sd.items()})
        res[mode] = rel(p32, e32.forward(inp)[1])
    print(name, "all k x k weights rounded:", {k: f"{v:.3e}" for k, v in res.items()})
