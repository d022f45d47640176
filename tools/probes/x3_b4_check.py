#!/usr/bin/env python3
"""Diagnostic: relative L1 of the depth map of the f32x3 engine / the float32-MFMA engine against the CPU oracle for several batch sizes and
zone-drop rates (found while writing tools/x3_two_term_probe.py: batch 4 with dropped zones read 1e-2)."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cfpnet_amd import spec, synthetic, weights, hip
from cfpnet_amd.engine import Engine
from oracle import cfpnet_oracle as O
layers = spec.COMBINE1_LAYERS
torch.set_num_threads(max(torch.get_num_threads(), 8))
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
lib = hip.load()
for B, drop, seed in ((4, 0.34, 4242), (4, 0.0, 4242), (8, 0.34, 4242), (2, 0.34, 4242), (4, 0.34, 97)):
    inp = synthetic.make_inputs(B, 480, 640, 8, 56, seed=seed, drop_hist=drop)
    p0 = O.forward(sd, inp, layer_names=layers)[1].numpy()
    dinp = synthetic.to_device(inp, "cuda:0")
    row = {"B": B, "drop": drop, "seed": seed}
    for name, kw, dbg in (("f32", dict(dtype=torch.float32), None), ("x3", dict(dtype=torch.float32, x3=True), None), ("x3_old_dw", dict(dtype=torch.float32, x3=True), (10, 0))):
        if dbg:
            lib.cfp_debug_set(*dbg)
        eng = Engine(sd, layer_names=layers, **kw)
        p1 = eng.forward(dinp)[1].cpu().numpy()
        if dbg:
            lib.cfp_debug_set(dbg[0], 1)
        row[name] = max(float(np.abs(p1[b] - p0[b]).sum() / np.abs(p0[b]).sum()) for b in range(B))
        del eng
    print(json.dumps(row), flush=True)
