#!/usr/bin/env python3
"""ms per batch of 8 in the benched mode and in latency mode with the library named by CFP_HIP_LIB (run it twice for an A/B of two builds)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
inp = synthetic.to_device(synthetic.make_inputs(8), "cuda:0")
out = []
for mode in (4, 1):
    best = 1e9
    for rnd in range(3):
        eng = Engine(sd, layer_names=layers, dtype=torch.bfloat16)
        eng.capture(inp, inflight=mode) if mode > 1 else eng.capture(inp)
        run = eng.replay_async if mode > 1 else eng.replay
        for _ in range(12):
            run()
        torch.cuda.synchronize()
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(32):
                run()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 32 * 1e3)
        del eng
    out.append(best)
print(f"{os.environ.get('CFP_HIP_LIB', 'default library')}: four in flight {out[0]:.3f} ms, one graph {out[1]:.3f} ms")
