#!/usr/bin/env python3
"""The launch list of one batch-8 forward in issue order with per-call device time (HIP events in an eager pass, median of 5):
which C-ABI call, which problem size, how long.  For finding launches to merge or hoist."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.engine import Engine
B = int(os.environ.get("B", "8"))
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
eng = Engine(sd, layer_names=layers, dtype=torch.bfloat16)
eng.use_side_stream = False
inp = synthetic.to_device(synthetic.make_inputs(B), "cuda:0")
for _ in range(2):
    eng.forward(inp)
torch.cuda.synchronize()
real = hip.call
runs = []
for rep in range(5):
    recs = []
    def timed(name, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); real(name, *a); e1.record()
        ints = [x for x in a if isinstance(x, int) and 0 < x < 10**7][:14]
        recs.append((name, e0, e1, ints))
    hip.call = timed
    eng.forward(inp)
    torch.cuda.synchronize()
    hip.call = real
    runs.append([(n, e0.elapsed_time(e1) * 1e3, ints) for n, e0, e1, ints in recs])
n = len(runs[0])
tot = 0.0
for i in range(n):
    ts = sorted(r[i][1] for r in runs)
    us = ts[2]
    tot += us
    print(f"{i:3d} {runs[0][i][0][4:]:28s} {us:7.1f} us   {runs[0][i][2]}")
print(f"{n} calls, sum {tot / 1e3:.3f} ms")
