#!/usr/bin/env python3
"""Graph-timed micro-benchmark of the large-kernel depthwise kernels (VALU vs banded-Toeplitz MFMA) on the LKPM shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
hip.load()
DEV = "cuda:0"
B = 8
for (H, W, C, k) in [(120, 160, 32, 31), (60, 80, 64, 15), (30, 40, 128, 7)]:
    x = ops.Act(torch.randn(B * H * W, C, device=DEV).to(torch.bfloat16), 0, C)
    out = ops.new_act(B * H * W, C, torch.bfloat16, DEV)
    w = torch.randn(C, 1, k, k) / k
    wa = w[:, 0].transpose(1, 2).reshape(C, k * k).contiguous().to(DEV)
    tb = ops.toeplitz_bands(w, torch.bfloat16).to(DEV)
    sc, sh = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    t_valu = graph_time_us(lambda: ops.dwconv_large(x, wa, sc, sh, out, B, H, W, k, hip.ACT_RELU), calls=6, replays=4)
    t_mfma = graph_time_us(lambda: ops.dwconv_large_mfma(x, tb, sc, sh, out, B, H, W, k, hip.ACT_RELU), calls=6, replays=4)
    fl = 2.0 * k * k * B * H * W * C
    print(f"{B}x{H}x{W}x{C} k{k}: VALU {t_valu:7.1f} us ({fl / t_valu / 1e6:6.1f} TFLOP/s)   Toeplitz-MFMA {t_mfma:7.1f} us ({fl / t_mfma / 1e6:6.1f} TFLOP/s useful)")
