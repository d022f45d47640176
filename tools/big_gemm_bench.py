#!/usr/bin/env python3
"""The largest implicit GEMMs of the forward (head 3x3 conv, decoder convs) under forced gen-2 tile variants."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
lib = hip.load()
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = [("head 3x3 128->128 @1/2", 240, 320, 128, 128, 3), ("conv0 3x3 32->128 @1/2", 240, 320, 32, 128, 3), ("up3.a 3x3 168->64 @1/4", 120, 160, 168, 64, 3),
          ("up2.a 3x3 312->128 @1/8", 60, 80, 312, 128, 3), ("up1.a 3x3 392->256 @1/16", 30, 40, 392, 256, 3), ("conv_out 1x1 128->256 @1/2", 240, 320, 128, 256, 1)]
for name, H, W, Cin, Cout, k in shapes:
    x = ops.Act(torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16), 0, Cin)
    w = (torch.randn(Cout, k * k * Cin, device=DEV) * 0.05).to(torch.bfloat16)
    out = ops.new_act(B * H * W, Cout, torch.bfloat16, DEV)
    sc, sh = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    fl = 2.0 * B * H * W * Cout * k * k * Cin
    def run():
        ops.conv2d(x, w, sc, sh, out, B, H, W, k, k, 1, k // 2, k // 2, H, W, hip.ACT_LRELU)
    res = []
    for v in (-1, 0, 1, 14, 15, 17):
        lib.cfp_debug_set(0, v)
        try:
            t = graph_time_us(run, calls=6, replays=4)
            res.append(f"v{v}: {t:7.1f} us {fl / t / 1e6:6.0f} TF/s")
        except Exception as e:
            res.append(f"v{v}: failed {str(e)[:40]}")
    lib.cfp_debug_set(0, -1)
    print(f"{name:28s} " + " | ".join(res))
