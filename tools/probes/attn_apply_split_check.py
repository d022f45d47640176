#!/usr/bin/env python3
"""cfp_attn_apply with the outputs of a (token, head) pair split over several lanes (cfp_debug_set key 38) against one lane per pair: EQUAL forward results."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.engine import Engine
lib = hip.load()
layers = spec.COMBINE1_LAYERS
for fam in ("uniform",):
    sd = weights.make_torch_state_dict(spec.model_manifest(layers), family=fam)
    eng = Engine(sd, layer_names=layers, dtype=torch.float32, x3=True)
    for B in (1, 2):
        inp = synthetic.to_device(synthetic.make_inputs(B, seed=3 + B), "cuda:0")
        outs = []
        for v in (0, 65536):
            lib.cfp_debug_set(38, v)
            o = eng.forward(inp, return_prob=True)
            torch.cuda.synchronize()
            outs.append([t.clone() for t in o if torch.is_tensor(t)])
        print(fam, B, all(torch.equal(a, b) for a, b in zip(*outs)), float(outs[0][1].std()))
lib.cfp_debug_set(38, 65536)
