#!/usr/bin/env python3
"""Where the chunk-pipelined f16x3 3x3 kernel's time goes (round 5): the head conv (614400 px, 128 -> 128) and a DAPM-type conv through
conv3x3_chunk_x3_kernel with parts switched off (cfp_debug_set key 16: 1 = no weight DMA after the prologue, 2 = no fragment reads / MFMAs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
lib = hip.load()
DEV = "cuda:0"
for (B, H, W, Cin, Cout, force) in ((8, 240, 320, 128, 128, 522), (8, 240, 320, 128, 128, 523), (8, 240, 320, 128, 128, 536), (8, 240, 320, 128, 128, 524),
                                    (8, 120, 160, 64, 64, 524), (8, 120, 160, 64, 64, 538), (8, 120, 160, 64, 64, 521), (8, 60, 80, 128, 64, 524), (8, 60, 80, 128, 64, 538),
                                    (8, 60, 80, 128, 128, 523), (8, 60, 80, 128, 128, 536), (1, 240, 320, 128, 128, 524), (1, 240, 320, 128, 128, 536)):
    x = ops.Act(torch.randn(B * H * W, Cin, device=DEV), 0, Cin)
    w = torch.randn(Cout, 9 * Cin, device=DEV) / (3 * Cin ** 0.5)
    wx = ops.pack_w_x3(w)
    out = ops.new_act(B * H * W, Cout, torch.float32, DEV)
    sc, sh = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    fn = lambda: ops.conv2d(x, wx, sc, sh, out, B, H, W, 3, 3, 1, 1, 1, H, W, hip.ACT_NONE)
    lib.cfp_debug_set(0, force)
    res = {}
    for name, pr in (("full", 0), ("no_dma", 1), ("no_compute", 2), ("neither", 3)):
        lib.cfp_debug_set(16, pr)
        res[name] = graph_time_us(fn, calls=6, replays=4)
    lib.cfp_debug_set(16, 0); lib.cfp_debug_set(0, -1)
    nks = 9 * Cin // 32
    print(f"{B}x{H}x{W} {Cin}->{Cout} variant {force}: " + "  ".join(f"{k} {v:7.1f} us" for k, v in res.items()) + f"   ({nks} K-steps)", flush=True)
