#!/usr/bin/env python3
"""Per-call table of one EAGER training step (416x544 crops, batch 16, 16-bit storage): HIP events around every C-ABI call, grouped by
(entry point, shape), with the algorithmic bytes / FLOPs of the convolution families -> where the step is far from its HBM roof.
Usage: python tools/train_calls.py [--dtype f16] [--top 60]"""
import argparse, collections, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.trainer import Trainer

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--dtype", default="f16"); ap.add_argument("--top", type=int, default=60)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
H, W = 416, 544
inp = synthetic.to_device(synthetic.make_inputs(a.batch, H, W, 6, 64, seed=5, drop_hist=0.34), "cuda:0")
target = torch.from_numpy(np.stack([synthetic.make_depth(H, W, seed=50 + i, holes=0.1) for i in range(a.batch)]))[:, None].cuda()
tr = Trainer(sd, layers, lr=3e-4, total_steps=100, dtype=DT)
for _ in range(2):
    tr.step(inp, target)
torch.cuda.synchronize()
real_call = hip.call
recs = []


def shape_of(name, args):
    if name in ("cfp_conv2d_nhwc", "cfp_conv2d_nhwc_ex", "cfp_conv2d_nhwc_moments"):
        o = 9 if name != "cfp_conv2d_nhwc_moments" else 6
        B, Hh, Ww, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo = args[o:o + 12]
        M = B * Ho * Wo
        return ("fwd", M, Cout, KH * KW * Cin, KH, stride), 2.0 * M * Cout * KH * KW * Cin, 2.0 * (M * Cout + B * Hh * Ww * Cin)
    if name == "cfp_conv2d_dgrad":
        B, Hh, Ww, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo = args[5:17]
        return ("dgrad", B * Hh * Ww, Cin, KH * KW * Cout, KH, stride), 2.0 * B * Hh * Ww * Cin * KH * KW * Cout / (stride * stride), 2.0 * (B * Ho * Wo * Cout + B * Hh * Ww * Cin)
    if name in ("cfp_conv2d_wgrad", "cfp_conv2d_wgrad_bias", "cfp_conv2d_wgrad_deferred"):
        o = 5 if name == "cfp_conv2d_wgrad" else 6
        B, Hh, Ww, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo = args[o:o + 12]
        return ("wgrad", B * Ho * Wo, Cout, KH * KW * Cin, KH, stride), 2.0 * B * Ho * Wo * Cout * KH * KW * Cin, 2.0 * (B * Ho * Wo * Cout + B * Hh * Ww * Cin)
    ints = tuple(int(x) for x in args if isinstance(x, int) and 0 < x < (1 << 31))[:6]
    return ints, 0.0, 0.0


def timed_call(name, *args):
    key, fl, by = shape_of(name, args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); real_call(name, *args); e1.record()
    recs.append((name, key, e0, e1, fl, by))


agg = collections.OrderedDict()
hip.call = timed_call
try:
    for r in range(a.reps):
        recs.clear()
        tr.net.zero_grad()
        tr._grads_to_flat(inp, target, tr.draw_pos_offsets(H, W))
        torch.cuda.synchronize()
        for name, key, e0, e1, fl, by in recs:
            d = agg.setdefault((name, key), {"n": 0, "ms": [0.0] * a.reps, "fl": fl, "by": by})
            d["ms"][r] += e0.elapsed_time(e1)
            if r == 0:
                d["n"] += 1
finally:
    hip.call = real_call
rows = []
for (name, key), d in agg.items():
    ms = min(d["ms"])
    rows.append((ms, name, key, d))
rows.sort(key=lambda x: -x[0])
tot = sum(r[0] for r in rows)
print(f"total {tot:.2f} ms over {sum(r[3]['n'] for r in rows)} calls (eager, event pairs; min of {a.reps} steps per group)")
byname = collections.defaultdict(lambda: [0, 0.0])
for ms, name, key, d in rows:
    byname[name][0] += d["n"]; byname[name][1] += ms
print("-- by entry point")
for name, (n, ms) in sorted(byname.items(), key=lambda x: -x[1][1])[:40]:
    print(f"{name:42s} {n:5d} {ms:8.3f} ms")
print("-- by (entry point, shape)")
for ms, name, key, d in rows[:a.top]:
    us = ms * 1e3 / d["n"]
    extra = f"  {d['by'] / us / 1e3:7.0f} GB/s {d['fl'] / us / 1e6:7.1f} TF/s" if d["by"] else ""
    print(f"{ms:7.3f} ms {d['n']:3d} x {us:7.1f} us  {name:34s} {str(key):46s}{extra}")
