#!/usr/bin/env python3
"""Where does the 16-bit error come from?  Runs the f32, the bf16 and the f16 engine on the same inputs and prints the relative
L1 difference of every tapped intermediate, in forward order."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine

layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
inp = synthetic.to_device(synthetic.make_inputs(1), "cuda:0")
res = {}
for dt in (torch.float32, torch.bfloat16, torch.float16):
    eng = Engine(sd, layer_names=layers, dtype=dt)
    taps = {}
    e, p, pr = eng.forward(inp, taps=taps)
    torch.cuda.synchronize()
    taps["pred"] = p.float().cpu()
    taps["edges"] = e.float().cpu()
    taps["prob"] = pr.float().cpu()
    res[dt] = taps
for k in res[torch.float32]:
    a = res[torch.float32][k].double().numpy()
    r = {dt: np.abs(a - res[dt][k].double().numpy()).sum() / max(np.abs(a).sum(), 1e-30) for dt in (torch.bfloat16, torch.float16)}
    print(f"{k:40s} relL1 bf16 {r[torch.bfloat16]:.3e}  f16 {r[torch.float16]:.3e}   max|f32| {np.abs(a).max():9.3f}")
