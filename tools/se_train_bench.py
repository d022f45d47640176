#!/usr/bin/env python3
"""Squeeze-excite of the training step at the encoder's shapes, piece by piece (us per launch, back-to-back in a HIP graph)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops, train_ops
from _gtime import graph_time_us
dev = "cuda:0"
for (B, HW, C, R) in [(16, 26 * 34, 816, 34), (16, 13 * 17, 1392, 58), (16, 26 * 34, 384, 16)]:
    Rp = -(-R // 4) * 4
    x = torch.randn(B * HW, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(B * HW, C, device=dev).to(torch.bfloat16)
    w1, b1 = torch.randn(Rp, C, device=dev) * 0.03, torch.zeros(Rp, device=dev)
    w2, b2 = torch.randn(C, Rp, device=dev) * 0.1, torch.zeros(C, device=dev)
    ns = max(1, min(64, HW // 16, -(-1024 // B)))
    part = torch.empty(B * ns, C, device=dev)
    xa = ops.Act(x, 0, C)
    ops.channel_sum(xa, part, B, HW, ns)
    mean, z1, gate = train_ops.se_train_fwd(part, ns, 1.0 / HW, w1, b1, w2, b2, B)
    dgate = train_ops.channel_dot(x, dy, B, HW)
    res = train_ops.se_train_bwd(dgate, gate, z1, mean, w1, w2, 1.0 / HW)
    t = {}
    t["channel_sum"] = graph_time_us(lambda: ops.channel_sum(xa, part, B, HW, ns))
    t["se_train_fwd"] = graph_time_us(lambda: train_ops.se_train_fwd(part, ns, 1.0 / HW, w1, b1, w2, b2, B))
    t["bcast_fma"] = graph_time_us(lambda: train_ops.bcast_fma(x, gate, None, B, HW))
    t["channel_dot"] = graph_time_us(lambda: train_ops.channel_dot(x, dy, B, HW))
    t["se_train_bwd(2 kernels)"] = graph_time_us(lambda: train_ops.se_train_bwd(dgate, gate, z1, mean, w1, w2, 1.0 / HW))
    t["bcast_fma+add"] = graph_time_us(lambda: train_ops.bcast_fma(dy, gate, res[4], B, HW))
    print(f"B={B} HW={HW} C={C} R={R}: " + "  ".join(f"{k} {v:.1f}" for k, v in t.items()) + f"  | sum {sum(t.values()):.1f} us")
