#!/usr/bin/env python3
"""How close are the 16-bit training gradients to the float32 ones at the real batch shape (16 x 416x544)?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.train_model import TrainNet
B, H, W = 16, 416, 544
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
inp = synthetic.to_device(synthetic.make_inputs(B, H, W, 6, 64, seed=5, drop_hist=0.34), "cuda:0")
target = torch.from_numpy(np.stack([synthetic.make_depth(H, W, seed=50 + i, holes=0.1) for i in range(B)]))[:, None].cuda()
offs = {"cross_atten3": (2, 3), "cross_atten2": (4, 7), "cross_atten1": (9, 11)}
res = {}
for dt in (torch.float32, torch.float16, torch.bfloat16):
    net = TrainNet(sd, layers, "cuda:0", dtype=dt)
    loss, _, _ = net.forward_backward(inp, target, target > 1e-3, pos_offsets=offs)
    g = net.grads()
    res[dt] = (float(loss), torch.cat([g[k].reshape(-1) for k in sorted(g)]).double())
    del net
    torch.cuda.empty_cache()
l32, g32 = res[torch.float32]
for dt in (torch.float16, torch.bfloat16):
    l, g = res[dt]
    cos = float((g * g32).sum() / (g.norm() * g32.norm()))
    print(f"{dt}: loss {l:.5f} (f32 {l32:.5f}), |g| ratio {float(g.norm() / g32.norm()):.4f}, cosine to the f32 gradient {cos:.4f}")
