#!/usr/bin/env python3
"""Every dense-conv launch of one training step (forward convs and data-gradient convs, bf16, 16 crops of 416x544) timed under
the automatic tile plan and under forced gen-2 tile variants: how much the plan fitted on the batch-8 inference shapes leaves
on the table for the training shapes."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.train_model import TrainNet
from _gtime import graph_time_us

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--out", default="gpurun_out/train_conv_sweep.json")
a = ap.parse_args()
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
H, W = 416, 544
inp = synthetic.to_device(synthetic.make_inputs(a.batch, H, W, 6, 64, seed=5, drop_hist=0.34), "cuda:0")
target = torch.from_numpy(np.stack([synthetic.make_depth(H, W, seed=50 + i, holes=0.1) for i in range(a.batch)]))[:, None].cuda()
net = TrainNet(sd, layers, "cuda:0", dtype=torch.bfloat16)
net.forward_backward(inp, target, target > 1e-3)
net.zero_grad()
lib = hip.load()
calls, keep = [], []
real = hip.call


def rec(name, *args):
    if name in ("cfp_conv2d_nhwc_ex", "cfp_conv2d_dgrad"):
        calls.append((name, args))
    real(name, *args)


hip.call = rec
net.forward_backward(inp, target, target > 1e-3)        # tensors stay referenced by the tape until zero_grad: pointers stay valid
torch.cuda.synchronize()
hip.call = real
from cfpnet_amd import ops, train_ops
uniq = {}
for name, args in calls:
    if name == "cfp_conv2d_nhwc_ex":
        B, Hh, Ww, Cin, Cout, KH, KW, st, pt, pl, Ho, Wo = args[9:21]
        if args[22] == 0:          # float32 few-row layers
            continue
        key = ("fwd", B * Ho * Wo, Cout, KH * KW * Cin, KH, st)
        geo = (B, Hh, Ww, Cin, Cout, KH, st, pt, pl, Ho, Wo)
    else:
        B, Hh, Ww, Cin, Cout, KH, KW, st, pt, pl, Ho, Wo = args[5:17]
        if st != 1 or args[18] == 0:
            continue               # float32 few-row layers; strided data gradients take the dilated first-generation kernel whatever the plan
        key = ("dgrad", B * Hh * Ww, Cin, KH * KW * Cout, KH, 1)
        geo = (B, Hh, Ww, Cin, Cout, KH, st, pt, pl, Ho, Wo)
    uniq.setdefault(key, [geo, 0])[1] += 1
del calls
net.zero_grad()
torch.cuda.synchronize()
print(len(uniq), "distinct 16-bit problems")
rows, tot_auto, tot_best = [], 0.0, 0.0
DEV, DT = "cuda:0", torch.bfloat16
for key, (geo, cnt) in uniq.items():
    B, Hh, Ww, Cin, Cout, KH, st, pt, pl, Ho, Wo = geo
    # the recorded pointers died with the tape: every problem gets its own buffers of the recorded shapes
    if key[0] == "fwd":
        x = ops.Act(torch.randn(B * Hh * Ww, Cin, device=DEV).to(DT), 0, Cin)
        w = (torch.randn(Cout, KH * KH * Cin, device=DEV) * 0.05).to(DT)
        out = ops.new_act(B * Ho * Wo, Cout, DT, DEV)
        fn = lambda: ops.conv2d(x, w, None, None, out, B, Hh, Ww, KH, KH, st, pt, pl, Ho, Wo)
    else:
        dy = torch.randn(B * Ho * Wo, Cout, device=DEV).to(DT)
        wt = (torch.randn(Cin, KH * KH * Cout, device=DEV) * 0.05).to(DT)
        dx = torch.empty(B * Hh * Ww, Cin, dtype=DT, device=DEV)
        fn = lambda: train_ops.conv2d_dgrad(dy, wt, B, Hh, Ww, Cin, KH, KH, st, pt, pl, Ho, Wo, dx=dx)
    fn()
    torch.cuda.synchronize()
    t_auto = graph_time_us(fn, calls=6, replays=3)
    best, sweep = (t_auto, "auto"), {}
    for v in (1, 2, 4, 8, 12, 13, 14, 15, 16):
        lib.cfp_debug_set(0, v)
        try:
            t = graph_time_us(fn, calls=6, replays=3)
        except Exception:
            continue
        finally:
            lib.cfp_debug_set(0, -1)
        sweep[v] = t
        if t < best[0]:
            best = (t, f"v{v}")
    kind, M, N, K, k, st = key
    rows.append(dict(kind=kind, M=M, N=N, K=K, k=k, count=cnt, auto_us=t_auto, best_us=best[0], best=best[1], sweep=sweep))
    tot_auto += cnt * t_auto; tot_best += cnt * best[0]
rows.sort(key=lambda r: -(r["auto_us"] - r["best_us"]) * r["count"])
for r in rows[:40]:
    print(f"{r['kind']:5s} M={r['M']:7d} N={r['N']:5d} K={r['K']:5d} k={r['k']} x{r['count']:2d} auto {r['auto_us']:7.1f} best {r['best_us']:7.1f} ({r['best']})  saves {(r['auto_us'] - r['best_us']) * r['count']:7.1f} us")
print(f"total per step: auto {tot_auto / 1e3:.3f} ms, best-of-sweep {tot_best / 1e3:.3f} ms")
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(rows, open(a.out, "w"))
