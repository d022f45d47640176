#!/usr/bin/env python3
"""Single-image long-K convolutions (3x3 at 1/16 and 1/8 scale, K = 576 ... 3528, 1 200 - 4 800 pixels): the automatic f16x3 plan (two K groups
in a workgroup, variant 19) against REAL K splits (slab workspace given) of the 64 x 64 / 32 x 64 tiles.  us per launch, graph-timed."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
DEV = "cuda:0"
lib = hip.load()
CASES = [(1, 30, 40, 392, 256, 3), (1, 60, 80, 312, 128, 3), (1, 30, 40, 256, 256, 3), (1, 60, 80, 128, 64, 3), (1, 60, 80, 128, 128, 3), (1, 60, 80, 64, 64, 3),
         (1, 30, 40, 128, 128, 3), (1, 30, 40, 512, 128, 1), (1, 15, 20, 1392, 232, 1),
         (1, 120, 160, 168, 64, 3), (1, 120, 160, 64, 64, 3), (1, 60, 80, 40, 160, 3), (1, 60, 80, 56, 224, 3)]
for B, H, W, Cin, Cout, k in CASES:
    M, K = B * H * W, k * k * Cin
    x = ops.Act(torch.randn(M, Cin, device=DEV), 0, Cin)
    w = ops.pack_w_x3((torch.randn(Cout, K, device=DEV) / math.sqrt(K)).contiguous())
    out = ops.new_act(M, Cout, torch.float32, DEV)
    ws = torch.empty(8 * M * Cout, device=DEV)
    sc, sh = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    fn = lambda: ops.conv2d(x, w, sc, sh, out, B, H, W, k, k, 1, k // 2, k // 2, H, W, hip.ACT_NONE, None, ws)
    res = {}
    lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1)
    fn(); torch.cuda.synchronize()
    res["auto"] = min(graph_time_us(fn, calls=12, replays=4) for _ in range(2))
    for v in (13, 17, 4):
        for sp in (2, 4, 8):
            lib.cfp_debug_set(0, 400 + v); lib.cfp_debug_set(1, sp)
            fn(); torch.cuda.synchronize()
            res[f"v{v}/s{sp}"] = min(graph_time_us(fn, calls=12, replays=4) for _ in range(2))
    lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1)
    best = min(res, key=res.get)
    print(f"{M:5d} x {Cout:4d} x {K:5d} (k{k}): auto {res['auto']:6.1f}  best {best} {res[best]:6.1f}   " + "  ".join(f"{k2}={v2:.1f}" for k2, v2 in res.items() if k2 != "auto"))
