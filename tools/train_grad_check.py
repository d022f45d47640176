#!/usr/bin/env python3
"""Is a gradient mismatch rounding noise or a bug?  Ground truth = autograd of the CPU oracle in float64; the float32
oracle and the HIP tape are both compared against it, tensor by tensor."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_train_step_gpu import _case
from oracle import cfpnet_oracle as O
from cfpnet_amd.train_model import TrainNet

import json
geom = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
layers, sd, inp, target, offs = _case(**geom)
torch.set_num_threads(8)


def step(dtype):
    sdg = {}
    for k, v in sd.items():
        v = v.detach().clone()
        if v.is_floating_point():
            v = v.to(dtype)
            if not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        sdg[k] = v
    cast = lambda x: x.to(dtype) if torch.is_tensor(x) and x.is_floating_point() else x
    i2 = {"rgb": cast(inp["rgb"]), "additional": {k: cast(v) for k, v in inp["additional"].items()}}
    O.BN_TRAIN = True
    taps = {}
    try:
        e, pred, prob = O.forward(sdg, i2, layer_names=layers, pos_offsets=offs, grad=True, taps=taps)
        for v in taps.values():
            if v.requires_grad:
                v.retain_grad()
        loss = O.silog_loss(pred, target.to(dtype), target > 1e-3)
        loss.backward()
    finally:
        O.BN_TRAIN = False
    TAPS[dtype] = {k: v.grad.double() for k, v in taps.items() if v.requires_grad and v.grad is not None}
    return float(loss), {k: v.grad.double() for k, v in sdg.items() if getattr(v, "grad", None) is not None}

TAPS = {}

l64, g64 = step(torch.float64)
l32, g32 = step(torch.float32)
net = TrainNet(sd, layers, "cuda:0")
net.record = {}
l1, _, _ = net.forward_backward(inp, target, target > 1e-3, pos_offsets=offs)
gh = {k: v.double().cpu() for k, v in net.grads().items()}
print("loss f64", l64, "f32", l32, "hip", float(l1))
gmax = max(float(g.abs().max()) for g in g64.values())
rows = []
for k, g in g64.items():
    if float(g.abs().max()) == 0 or k not in gh:
        continue
    den = max(float(g.abs().max()), 1e-5 * gmax)
    rows.append((float((gh[k] - g).abs().max()) / den, float((g32[k] - g).abs().max()) / den, k))
rows.sort(reverse=True)
print(f"{'hip vs f64':>12} {'f32 oracle vs f64':>18}  tensor")
for a, b, k in rows[:25]:
    print(f"{a:12.3e} {b:18.3e}  {k}")
print("median hip", np.median([r[0] for r in rows]), "median f32 oracle", np.median([r[1] for r in rows]))
print("tensors where hip error > 5x the f32 oracle's and > 1e-3:", [(f"{a:.1e}", f"{b:.1e}", k) for a, b, k in rows if a > 5 * b and a > 1e-3][:20])

for a, b, k in rows[:6]:
    e = (gh[k] - g64[k]).abs(); den = float(g64[k].abs().max())
    fro = float((gh[k] - g64[k]).norm() / g64[k].norm()); fro32 = float((g32[k] - g64[k]).norm() / g64[k].norm())
    idx = int(e.reshape(-1).argmax())
    print(f"{k}: shape {tuple(g64[k].shape)} fro-rel hip {fro:.2e} f32 {fro32:.2e}; elements with err > 1e-3*max: {int((e > 1e-3 * den).sum())}/{e.numel()}; argmax {np.unravel_index(idx, tuple(g64[k].shape))}")

print("activation gradients at the fusion layer outputs (relative to the f64 tensor's max): hip | f32 oracle")
for k in sorted(net.record):
    if k not in TAPS[torch.float64]:
        continue
    g64t = TAPS[torch.float64][k]; g32t = TAPS[torch.float32][k]
    v = net.record[k]
    if v.g is None:
        continue
    gh_ = v.g.double().cpu().reshape(g64t.shape)
    den = float(g64t.abs().max())
    print(f"  {k:44s} {float((gh_ - g64t).abs().max()) / den:.2e} | {float((g32t - g64t).abs().max()) / den:.2e}")
