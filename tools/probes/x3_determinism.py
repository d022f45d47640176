#!/usr/bin/env python3
"""Is the f32x3 forward deterministic (eager twice, eager vs captured replay)?  Bisect by switching single kernels off."""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.engine import Engine
lib = hip.load()
base = (640, 960) if "--config5" in sys.argv else (480, 640)
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers, base_resolution=base))
inp = synthetic.to_device(synthetic.make_inputs(2, base[0], base[1], 16 if base[0] == 640 else 8, 40 if base[0] == 640 else 56, seed=31, drop_hist=0.2, image_hw=base), "cuda:0")


def run(tag, env=(), knobs=()):
    for k, v in env:
        os.environ[k] = v
    for k, v in knobs:
        lib.cfp_debug_set(k, v)
    try:
        eng = Engine(sd, layer_names=layers, base_resolution=base)
        t1, t2 = {}, {}
        _, a, pa = eng.forward(inp, taps=t1)
        _, b, pb = eng.forward(inp, taps=t2)
        eng.capture(inp)
        _, c, pc = eng.replay()
        _, d, _ = eng.replay()
        torch.cuda.synchronize()
        bad = [k for k in t1 if not torch.equal(t1[k], t2[k])]
        print(f"{tag:40s} eager==eager {torch.equal(a, b)}  eager==replay {torch.equal(a, c)}  replay==replay {torch.equal(c, d)}  prob {torch.equal(pa, pb)} {torch.equal(pa, pc)}"
              f"  max|a-c| {float((a - c).abs().max()):.3e}  first differing taps (eager pair): {bad[:4]}")
    finally:
        for k, v in env:
            os.environ.pop(k, None)
        for k, v in knobs:
            lib.cfp_debug_set(k, 0 if k != 24 else 1)


run("default")
run("plain K loop (28=1)", knobs=((28, 1),))
run("no halo / chunk (24=0)", knobs=((24, 0),))
run("CFP_X3_TAIL=0", env=(("CFP_X3_TAIL", "0"),))
run("CFP_X3_DWLARGE=0", env=(("CFP_X3_DWLARGE", "0"),))
run("CFP_BIN_HEAD_FUSED=0", env=(("CFP_BIN_HEAD_FUSED", "0"),))
run("no LN fuse (26=0)", knobs=((26, 0),))
run("CFP_LKPM_FUSED=0", env=(("CFP_LKPM_FUSED", "0"),))
run("CFP_X2I_HOIST=0", env=(("CFP_X2I_HOIST", "0"),))
run("CFP_TAIL_Q=0", env=(("CFP_TAIL_Q", "0"),))
run("CFP_NO_SIDE_STREAM=1", env=(("CFP_NO_SIDE_STREAM", "1"),))
