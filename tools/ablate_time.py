#!/usr/bin/env python3
"""What each kernel family costs IN THE BENCHED MODE (four batches in flight), not in isolation: the step is re-captured with one
family's launches skipped (results are garbage, timing is not) and the drop of ms/step is that family's marginal cost.  Families that
fill the chip cost their isolated time; small launches hide behind the other batches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops, spec, synthetic, weights
from cfpnet_amd.engine import Engine

X3 = "--x3" in sys.argv          # the default boundary mode (float32 storage, f16x3 matrix math) instead of bf16
ENG_KW = dict(dtype=torch.float32, x3=True) if X3 else dict(dtype=torch.bfloat16)
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
inp = synthetic.to_device(synthetic.make_inputs(8), "cuda:0")
real = hip.call


def conv_family(a):
    B, H, W, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo = a[9:21]
    x3 = len(a) > 26 and (a[26] & hip.CONV_X3)
    v, sp = ops.conv2d_plan(B * Ho * Wo, Cout, KH * KW * Cin, hip.F32X3 if x3 else a[22], 0, B, KH, stride)
    big = B * Ho * Wo * Cout >= 153600 * 64
    if x3:
        return "conv:halo_x3" if v >= 500 else ("conv:igemm_x3-3x3" if KH == 3 else "conv:igemm_x3-1x1-big" if big else "conv:igemm_x3-1x1-small")
    return ("conv:halo" if v >= 300 else "conv:direct" if v >= 200 else "conv:igemm2-big" if (v >= 100 and big) else "conv:igemm2-small" if v >= 100 else
            "conv:gen1")


def family(name, a):
    if name in ("cfp_conv2d_nhwc", "cfp_conv2d_nhwc_ex"):
        return conv_family(a)
    if name.startswith("cfp_dwconv3x3"):
        return "dw3x3"
    return name[4:]


def measure(skip, inflight=4, reps=24):
    def call(name, *a):
        if family(name, a) in skip:
            return 0
        return real(name, *a)
    eng = Engine(sd, layer_names=layers, **ENG_KW)
    hip.call = call
    try:
        eng.capture(inp, inflight=inflight) if inflight > 1 else eng.capture(inp)
    finally:
        hip.call = real
    run = eng.replay_async if inflight > 1 else eng.replay
    for _ in range(12):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
    del eng
    return min(ts)


fams = {}
def rec(name, *a):
    fams[family(name, a)] = fams.get(family(name, a), 0) + 1
    return real(name, *a)
e0 = Engine(sd, layer_names=layers, **ENG_KW)
e0.forward(inp); torch.cuda.synchronize()
hip.call = rec
e0.forward(inp); torch.cuda.synchronize()
hip.call = real
del e0
for mode in ((4,) if X3 else (4, 1)):
    base = measure(set(), mode)
    print(f"--- {'four batches in flight' if mode > 1 else 'one graph per batch (latency mode)'}: baseline {base:.3f} ms per batch of 8")
    rows = []
    for f in sorted(fams):
        t = measure({f}, mode)
        rows.append((base - t, f, fams[f]))
    for d, f, n in sorted(rows, reverse=True):
        print(f"{f:36s} {n:3d} launches   marginal {d * 1e3:7.1f} us  = {100 * d / base:5.1f} %")
