#!/usr/bin/env python3
"""Per-launch timing table of one forward (HIP events around every C-ABI call)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops, spec, synthetic, weights
from cfpnet_amd.engine import Engine

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--top", type=int, default=60)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--group", action="store_true")
a = ap.parse_args()
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
eng = Engine(sd, layer_names=layers, dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32)
inp = synthetic.to_device(synthetic.make_inputs(a.batch), "cuda:0")
for _ in range(2): eng.forward(inp)
recs = []
real = hip.call
def timed(name, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); real(name, *args); e1.record()
    recs.append((name, args, e0, e1))
hip.call = timed
acc = {}
R = 5
for r in range(R):
    recs.clear(); eng.forward(inp); torch.cuda.synchronize()
    for i, (name, args, e0, e1) in enumerate(recs):
        acc.setdefault(i, [name, args, 0.0])[2] += e0.elapsed_time(e1) / R
rows = []
for i, (name, args, ms) in acc.items():
    desc, fl = "", 0
    if name in ("cfp_conv2d_nhwc", "cfp_conv2d_nhwc_ex"):
        B, H, W, Cin, Cout, KH, KW, st, pt, pl, Ho, Wo = args[9:21]
        M = B * Ho * Wo; fl = 2.0 * M * Cout * KH * KW * Cin
        ex = name.endswith("_ex")
        v, sp = ops.conv2d_plan(M, Cout, KH * KW * Cin, args[22], Ho * Wo if (ex and args[26]) else 0, B, KH, st)
        desc = f"M={M} N={Cout} K={KH*KW*Cin} k{KH} s{st} v{v}/s{sp}" + (" LN" if ex and args[23] else "") + (" PIW" if ex and args[26] else "")
    elif name == "cfp_dwconv3x3_nhwc":
        B, H, W, C, st = args[7:12]; desc = f"{B}x{H}x{W}x{C} s{st}"
    elif name == "cfp_dwconv_large_nhwc":
        B, H, W, C, k = args[7:12]; desc = f"{B}x{H}x{W}x{C} k{k}"; fl = 2.0 * k * k * B * H * W * C
    elif name in ("cfp_attn_kv_reduce",):
        desc = f"NB={args[7]} Hk={args[8]} Wk={args[9]} th={args[10]} tw={args[11]} h={args[18]} d={args[19]}"
    elif name == "cfp_attn_apply":
        desc = f"NB={args[6]} {args[7]}x{args[8]} h={args[17]} d={args[18]}"
    elif name == "cfp_se_hidden":
        desc = f"B={args[6]} C={args[7]} R={args[8]} ns={args[1]}"
    elif name == "cfp_se_scale":
        desc = f"B={args[5]} HW={args[6]} C={args[7]} R={args[8]}"
    elif name == "cfp_channel_sum":
        desc = f"B={args[3]} HW={args[4]} C={args[5]} ns={args[6]}"
    elif name == "cfp_layernorm":
        desc = f"rows={args[9]} C={args[10]}"
    elif name == "cfp_resize_bilinear":
        desc = f"{args[2]}x{args[3]} -> {args[14]}x{args[15]} C={args[22]}"
    rows.append((ms, i, name, desc, fl))
if a.group:
    g = {}
    for ms, i, name, desc, fl in rows:
        d = g.setdefault((name, desc), [0, 0.0, 0.0]); d[0] += 1; d[1] += ms; d[2] += fl
    tot = sum(r[0] for r in rows)
    print(f"launches {len(rows)} total {tot:.3f} ms")
    for (name, desc), (n, ms, fl) in sorted(g.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"{ms*1e3:9.1f} us  x{n:3d} ({ms/n*1e3:7.1f} each) {name:24s} {desc:52s} {fl/ms/1e9 if fl else 0:8.1f} TF/s")
    sys.exit(0)
tot = sum(r[0] for r in rows)
print(f"launches {len(rows)} total {tot:.3f} ms")
for ms, i, name, desc, fl in sorted(rows, reverse=True)[: a.top]:
    print(f"{ms*1e3:9.1f} us  #{i:3d} {name:24s} {desc:50s} {fl/ms/1e9 if fl else 0:8.1f} TF/s")
