#!/usr/bin/env python3
"""k = 31 large-kernel depthwise conv in the f16x3 mode: the 64 x 32 pixel tile (120 KB of LDS, one workgroup per CU) against the 32 x 32 tile
(79 KB, two per CU); batch 8 and 1 at the 1/4 scale (120 x 160 x 32 channels), alone and four copies side by side."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops
from cfpnet_amd.engine import concurrent_streams
from _gtime import graph_time_us_concurrent, graph_time_us
lib = hip.load()
DEV = "cuda:0"
ST = concurrent_streams(DEV, want=4)
for B in (8, 1):
    H, W, C, k = 120, 160, 32, 31
    x = ops.Act(torch.randn(B * H * W, C, device=DEV), 0, C)
    w = torch.randn(C, 1, k, k) / k
    tb = ops.toeplitz_bands_x3(w).to(DEV)
    sc, sh = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    out = ops.new_act(B * H * W, C, torch.float32, DEV)
    fn = lambda: ops.dwconv_large_mfma(x, tb, sc, sh, out, B, H, W, k, hip.ACT_RELU)
    for v in (0, 1):
        lib.cfp_debug_set(30, v)
        print(f"B={B} tile {'32x32' if v else '64x32'}: {graph_time_us(fn, calls=6, replays=4):7.1f} us alone, {graph_time_us_concurrent(fn, ST, calls=6, replays=3):7.1f} us per call with four copies", flush=True)
