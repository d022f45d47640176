#!/usr/bin/env python3
"""Stress of the in-flight (throughput) mode: N different inputs through `replay_async` with k slots, every result compared bit for
bit with the eager forward of the same input computed before AND after.  Tells which side is unstable."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine

from cfpnet_amd import hip
if os.environ.get("PROBE_V1") == "1":
    hip.load().cfp_debug_set(2, 1)          # gen-1 GEMM kernels only (register-staged loads, no LDS-DMA)
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
NI = 8
inps = [synthetic.to_device(synthetic.make_inputs(2, 256, 320, 3, 64, seed=40 + i, drop_hist=0.2 * (i % 2)), "cuda:0") for i in range(NI)]
dt = torch.float16
for slots in [int(v) for v in os.environ.get('PROBE_SLOTS', '3,4').split(',')]:
    eng = Engine(sd, layer_names=layers, dtype=dt)
    want = []
    for x in inps:
        want.append(eng.forward(x)[1].clone())
    torch.cuda.synchronize()
    again = [eng.forward(x)[1].clone() for x in inps]
    torch.cuda.synchronize()
    print(f"slots={slots}: eager vs eager again identical: {[bool(torch.equal(a, b)) for a, b in zip(want, again)]}")
    eng.capture(inps[0], inflight=slots)
    n = len(eng._slots)
    bad = 0
    rounds = int(os.environ.get('PROBE_ROUNDS', '100'))
    for r in range(rounds):
        got = []
        for i, x in enumerate(inps):
            (e, p, pr), ev = eng.replay_async(x)
            got.append((p, ev))
            if len(got) >= n:
                j = len(got) - n
                got[j][1].synchronize()
                got[j] = (got[j][0].clone(), None)
        torch.cuda.synchronize()
        for i, (p, _) in enumerate(got):
            if not torch.equal(p, want[i]):
                bad += 1
                d = (p - want[i]).abs()
                rows = torch.nonzero(d.flatten(2).amax(2).flatten() > 0).flatten().tolist()
                yy = torch.nonzero(d[0, 0].amax(1) > 0).flatten()
                if bad <= 6: print(f"  round {r} input {i} (slot {i % n}): mismatch, max |d| {float(d.max()):.2e}, images {rows}, rows of image 0: {yy[:3].tolist()}..{yy[-3:].tolist() if len(yy) else []}")
    print(f"slots={slots} (got {n}): {bad} mismatching results of {rounds * NI}", flush=True)
    del eng
