#!/usr/bin/env python3
"""Does the result of a forward depend on what the engine ran BEFORE it (a kernel reading a buffer before this forward wrote it)?
forward(B) after forward(A) vs forward(B) after forward(B), fresh engines, buffer by buffer."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine

layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
B, H, W, zn, zp = 2, 256, 320, 3, 64
A = synthetic.to_device(synthetic.make_inputs(B, H, W, zn, zp, seed=41, drop_hist=0.2), "cuda:0")
Bi = synthetic.to_device(synthetic.make_inputs(B, H, W, zn, zp, seed=42, drop_hist=0.0), "cuda:0")


def snap(eng):
    out = {}
    for k, v in eng._plans[(B, H, W, 0)]["bufs"].items():
        t = v.buf if hasattr(v, "buf") else v
        out[k] = t.clone()
    return out


for dt in (torch.float16, torch.bfloat16, torch.float32):
    res = []
    for first in (A, Bi):
        eng = Engine(sd, layer_names=layers, dtype=dt)
        eng.forward(first)
        torch.cuda.synchronize()
        e, p, pr = eng.forward(Bi)
        torch.cuda.synchronize()
        s = snap(eng)
        s["__pred"], s["__edges"], s["__prob"] = p.clone(), e.clone(), pr.clone()
        res.append(s)
        del eng
    diff = [k for k in res[0] if not torch.equal(res[0][k].view(torch.uint8), res[1][k].view(torch.uint8))]
    print(dt, "buffers that differ between histories:", diff if diff else "none", flush=True)
