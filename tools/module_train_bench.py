#!/usr/bin/env python3
"""Step time of the DROP-IN training path: `Deltar` in .train() under torch autograd with a torch optimizer, i.e. the
reference's own loop shape (train.py:104-135) -- forward / backward are the HIP tape, the optimizer is torch's."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import spec, synthetic, train_ops
from cfpnet_amd.deltar import Deltar

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dtype", default="bf16", choices=("f32", "bf16", "f16"))
a = ap.parse_args()
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
H, W = 416, 544
model = Deltar(n_bins=256, min_val=1e-3, max_val=10.0, dtype=DT).cuda().train()
opt = torch.optim.AdamW([{"params": list(model.get_1x_lr_params()), "lr": 3e-5}, {"params": list(model.get_10x_lr_params()), "lr": 3e-4}], weight_decay=0.1)
inp = synthetic.to_device(synthetic.make_inputs(a.batch, H, W, 6, 64, seed=5, drop_hist=0.34), "cuda:0")
target = torch.from_numpy(np.stack([synthetic.make_depth(H, W, seed=50 + i, holes=0.1) for i in range(a.batch)]))[:, None].cuda()
crit = train_ops.SILogLoss()


def step():
    opt.zero_grad()
    edges, pred = model(inp)
    p = torch.nn.functional.interpolate(pred, target.shape[-2:], mode="bilinear", align_corners=True)
    m = target > 1e-3
    g = torch.log(p[m]) - torch.log(target[m])
    loss = 10 * torch.sqrt(torch.var(g) + 0.15 * torch.pow(torch.mean(g), 2))       # loss.py:9-19 in torch, like the reference's criterion
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(json.dumps(dict(batch=a.batch, ms_per_step=dt * 1e3, samples_per_s=a.batch / dt, loss=float(loss), dtype=a.dtype, path="Deltar.train() + torch autograd + torch AdamW")))
