#!/usr/bin/env python3
"""Chunk-kernel variants with FOUR copies side by side (the benched mode) on the 3x3 problems the plan sends (or could send) to them; 4xx = implicit GEMM."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops
from cfpnet_amd.engine import concurrent_streams
from _gtime import graph_time_us_concurrent, graph_time_us
lib = hip.load()
DEV = "cuda:0"
ST = concurrent_streams(DEV, want=4)
ops.PLAN_IN_FLIGHT = True
for (B, H, W, Cin, Cout, forces) in ((8, 240, 320, 96, 32, (538, 544, 545, 546, -1)), (8, 120, 160, 192, 64, (538, 544, -1)), (8, 120, 160, 64, 64, (538, 544, -1)), (8, 120, 160, 64, 32, (538, 544, 545, 546, -1)),
                                     (8, 60, 80, 128, 64, (538, 544, -1)), (8, 60, 80, 64, 64, (538, 544, -1)), (8, 60, 80, 320, 128, (536, -1)), (8, 30, 40, 416, 256, (536, -1)), (1, 240, 320, 96, 32, (524, 538, 544, 545, -1)), (1, 120, 160, 192, 64, (538, 544, -1))):
    x = ops.Act(torch.randn(B * H * W, Cin, device=DEV), 0, Cin)
    w = torch.randn(Cout, 9 * Cin, device=DEV) / (3 * Cin ** 0.5)
    wx = ops.pack_w_x3(w)
    out = ops.new_act(B * H * W, Cout, torch.float32, DEV)
    sc, sh = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    fn = lambda: ops.conv2d(x, wx, sc, sh, out, B, H, W, 3, 3, 1, 1, 1, H, W, hip.ACT_NONE)
    res = []
    for f in forces:
        lib.cfp_debug_set(0, f)
        try:
            t4 = graph_time_us_concurrent(fn, ST, calls=6, replays=3)
            t1 = graph_time_us(fn, calls=6, replays=3)
            res.append(f"{'plan' if f < 0 else f}: {t4:6.1f} ({t1:6.1f} alone)")
        except RuntimeError as e:
            res.append(f"{f}: n/a")
    lib.cfp_debug_set(0, -1)
    print(f"{B}x{H}x{W} {Cin:3d}->{Cout:3d}: " + "   ".join(res), flush=True)
