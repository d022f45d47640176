#!/usr/bin/env python3
"""The squeeze-excite project GEMMs of a single image (per-image weights; 300 ... 4 800 rows, K = 224 ... 3 072): the automatic plan (K splits + reduce
launch where few tiles meet a long K) against un-split 32 x 64 / 64 x 64 / two-K-group tiles and other split counts.  us per launch (+ reduce), graph-timed."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
DEV = "cuda:0"
lib = hip.load()
CASES = [(300, 1392, 232), (300, 816, 232), (1200, 816, 136), (1200, 672, 136), (1200, 448, 112), (1200, 224, 112), (4800, 224, 56), (300, 2304, 384), (300, 1392, 384)]
for hw, K, Cout in CASES:
    x = ops.Act(torch.randn(hw, K, device=DEV), 0, K)
    w = ops.pack_w_x3((torch.randn(Cout, K, device=DEV) / math.sqrt(K)).contiguous()).unsqueeze(0).contiguous()
    out = ops.new_act(hw, Cout, torch.float32, DEV)
    ws = torch.empty(8 * hw * Cout, device=DEV)
    sc, sh = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    fn = lambda: ops.conv2d(x, w, sc, sh, out, 1, hw, 1, 1, 1, 1, 0, 0, hw, 1, hip.ACT_NONE, None, ws, per_image_weights=True)
    res = {}
    lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1)
    fn(); torch.cuda.synchronize()
    res["auto"] = min(graph_time_us(fn, calls=12, replays=4) for _ in range(2))
    for v, sp in ((17, 1), (13, 1), (19, 1), (17, 2), (17, 4), (13, 2), (13, 4), (13, 8), (4, 4), (4, 8)):
        lib.cfp_debug_set(0, 400 + v); lib.cfp_debug_set(1, sp)
        try:
            fn(); torch.cuda.synchronize()
            res[f"v{v}/s{sp}"] = min(graph_time_us(fn, calls=12, replays=4) for _ in range(2))
        except RuntimeError:
            pass
    lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1)
    pv, ps = ops.conv2d_plan(hw, Cout, K, hip.F32X3, hw, 1, 1, 1)
    best = min(res, key=res.get)
    print(f"{hw:5d} x {Cout:4d} x {K:5d}: plan v{pv - 400}/s{ps} auto {res['auto']:6.1f}  best {best} {res[best]:6.1f}   " + "  ".join(f"{k2}={v2:.1f}" for k2, v2 in res.items() if k2 != "auto"))
