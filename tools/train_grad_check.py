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

layers, sd, inp, target, offs = _case()
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
    try:
        e, pred, prob = O.forward(sdg, i2, layer_names=layers, pos_offsets=offs, grad=True)
        loss = O.silog_loss(pred, target.to(dtype), target > 1e-3)
        loss.backward()
    finally:
        O.BN_TRAIN = False
    return float(loss), {k: v.grad.double() for k, v in sdg.items() if getattr(v, "grad", None) is not None}

l64, g64 = step(torch.float64)
l32, g32 = step(torch.float32)
net = TrainNet(sd, layers, "cuda:0")
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
