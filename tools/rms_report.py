#!/usr/bin/env python3
"""Print per-stage activation statistics of the oracle forward with the synthetic parameters
(used to tune the init gains in cfpnet_amd/weights.py so the random network stays O(1))."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import spec, synthetic, weights
from oracle import cfpnet_oracle as O

def main():
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    inp = synthetic.make_inputs(1)
    taps = {}
    t = time.time()
    edges, pred, prob = O.forward(sd, inp, layer_names=layers, taps=taps)
    print(f"forward {time.time()-t:.2f}s")
    for k, v in taps.items():
        v = v.double()
        print(f"{k:42s} {tuple(v.shape)!s:24s} mean {float(v.mean()):9.4f} rms {float((v*v).mean().sqrt()):9.4f} max {float(v.abs().max()):9.3f}")
    print("pred mean/std", float(pred.mean()), float(pred.std()), "prob max mean", float(prob.max(1)[0].mean()))
    print("edges", edges[0, :4].tolist(), edges[0, -2:].tolist())

if __name__ == "__main__":
    main()
