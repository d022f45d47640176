#!/usr/bin/env python3
"""Run-to-run determinism of the eager forward and of the captured graphs, buffer by buffer: every activation buffer of the plan is
checksummed after each forward of the SAME input; a buffer whose bits change names the kernel to look at."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine

layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))


def sums(eng, key):
    out = {}
    for k, v in eng._plans[key]["bufs"].items():
        t = v.buf if hasattr(v, "buf") else v
        out[k] = int(t.contiguous().view(torch.uint8).to(torch.int64).sum())
    return out


for (B, H, W, zn, zp) in ((2, 256, 320, 3, 64), (8, 480, 640, 8, 56)):
    inp = synthetic.to_device(synthetic.make_inputs(B, H, W, zn, zp, seed=40, drop_hist=0.2), "cuda:0")
    for dt in (torch.float16, torch.bfloat16):
        eng = Engine(sd, layer_names=layers, dtype=dt)
        ref = None
        bad = {}
        n = 40 if B == 2 else 12
        for it in range(n):
            e, p, pr = eng.forward(inp)
            torch.cuda.synchronize()
            s = sums(eng, (B, H, W, 0))
            s["__pred"] = int(p.view(torch.uint8).to(torch.int64).sum()); s["__edges"] = int(e.view(torch.uint8).to(torch.int64).sum())
            if ref is None:
                ref = s
            else:
                for k in s:
                    if s[k] != ref[k]:
                        bad.setdefault(k, []).append(it)
        print(f"B={B} {H}x{W} {dt}: {n} eager forwards, buffers that changed: {({k: v[:4] for k, v in bad.items()} if bad else 'none')}", flush=True)
        del eng
