#!/usr/bin/env python3
"""A/B of one cfp_debug_set switch in the benched mode and in latency mode, same process, alternating:
    python tools/ab_debug_key.py 15 0 1        # key 15 with values 0 and 1"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.engine import Engine

key, vals = int(sys.argv[1]), [int(v) for v in sys.argv[2:]]
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
inp = synthetic.to_device(synthetic.make_inputs(8), "cuda:0")
lib = hip.load()


def measure(inflight, reps=32):
    eng = Engine(sd, layer_names=layers, dtype=torch.bfloat16)
    eng.capture(inp, inflight=inflight) if inflight > 1 else eng.capture(inp)
    run = eng.replay_async if inflight > 1 else eng.replay
    for _ in range(12):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
    del eng
    return min(ts)


res = {(v, m): [] for v in vals for m in (4, 1)}
for rnd in range(3):
    for v in vals:
        lib.cfp_debug_set(key, v)
        for m in (4, 1):
            res[(v, m)].append(measure(m))
for v in vals:
    print(f"key {key} = {v}: four in flight {min(res[(v, 4)]):.3f} ms ({', '.join(f'{t:.3f}' for t in res[(v, 4)])})   one graph {min(res[(v, 1)]):.3f} ms")
