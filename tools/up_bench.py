#!/usr/bin/env python3
"""cfp_upsample_cat_conv3x3 against cfp_resize_bilinear + cfp_conv2d_nhwc on the decoder's four UpSampleBN stages (batch 8),
back-to-back in a replayed HIP graph."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
DEV = "cuda:0"
B = int(os.environ.get("B", "8"))
for name, (Hs, Ws, H, W, Cup, Cskip, Cout) in {"up1.a": (15, 20, 30, 40, 256, 136, 256), "up2.a": (30, 40, 60, 80, 256, 56, 128),
                                               "up3.a": (60, 80, 120, 160, 128, 40, 64), "up4.a": (120, 160, 240, 320, 64, 16, 32)}.items():
    dt = torch.bfloat16
    Cin = Cup + Cskip
    low = ops.Act(torch.randn(B * Hs * Ws, Cup, device=DEV).to(dt), 0, Cup)
    cat = ops.new_act(B * H * W, Cin, dt, DEV); cat.buf.normal_()
    w = (torch.randn(Cout, 9 * Cin, device=DEV) / math.sqrt(9 * Cin)).to(dt)
    sc, sh = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    o1, o2 = ops.new_act(B * H * W, Cout, dt, DEV), ops.new_act(B * H * W, Cout, dt, DEV)
    fused = lambda: ops.upsample_cat_conv3x3(low, Hs, Ws, cat.slice(Cup, Cskip), w, sc, sh, o1, B, H, W, hip.ACT_LRELU)
    rs = lambda: ops.resize_bilinear(low, Hs, Ws, (0, 0, Hs, Ws), cat.slice(0, Cup), H, W, (0, 0, H, W), B)
    cv = lambda: ops.conv2d(cat, w, sc, sh, o2, B, H, W, 3, 3, 1, 1, 1, H, W, hip.ACT_LRELU)
    def pair():
        rs(); cv()
    tf, tr, tc, tp = (graph_time_us(f, calls=8, replays=4) for f in (fused, rs, cv, pair))
    lib = hip.load()
    lib.cfp_debug_set(14, 0)
    td = graph_time_us(fused, calls=8, replays=4)
    lib.cfp_debug_set(14, 2)
    extra = f"   fused through the direct kernel {td:7.1f}"
    if Cin <= 128 and Cout <= 64:
        for v in (0, 1, 2, 3, 7):
            if Cout > (16, 32, 64, 64, 0, 0, 0, 32)[v]:
                continue
            lib.cfp_debug_set(0, 300 + v)
            extra += f"  halo h{v} {graph_time_us(fused, calls=8, replays=4):6.1f}"
        lib.cfp_debug_set(0, -1)
    lib.cfp_debug_set(14, 1)
    print(f"{name}: fused {tf:7.1f} us   resize {tr:6.1f} + conv {tc:7.1f} = pair {tp:7.1f} us   ({2e-6 * B * H * W * Cout * 9 * Cin / tf:6.1f} TFLOP/s fused){extra}", flush=True)
