#!/usr/bin/env python3
"""cfp_dwconv_large_mfma_nhwc (LKPM's 31 / 15 / 7 depthwise kernels) at the benched shapes, graph-timed."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
DEV = "cuda:0"
dt = torch.bfloat16
for B, H, W, C, k in [(8, 120, 160, 32, 31), (8, 60, 80, 64, 15), (8, 30, 40, 128, 7)]:
    x = ops.new_act(B * H * W, C, dt, DEV); x.buf.normal_()
    out = ops.new_act(B * H * W, C, dt, DEV)
    w = torch.randn(C, k, k) * 0.05
    tb = ops.toeplitz_bands(w, dt).to(DEV)
    sc, sh = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    fn = lambda: ops.dwconv_large_mfma(x, tb, sc, sh, out, B, H, W, k, hip.ACT_RELU)
    fn(); torch.cuda.synchronize()
    t = min(graph_time_us(fn, calls=16, replays=5) for _ in range(2))
    print(f"k={k:2d} {B}x{H}x{W}x{C}: {t:6.1f} us per launch")
