#!/usr/bin/env python3
"""Inverted-residual front half at the encoder's shapes (batch 8): pointwise expand GEMM + depthwise 3x3 kernel (two launches, the
expanded tensor through HBM) against the fused kernel (csrc/mbconv.hip); back-to-back inside a replayed HIP graph."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from cfpnet_amd.engine import concurrent_streams
from _gtime import graph_time_us, graph_time_us_concurrent
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = concurrent_streams(DEV, 4)
for dt in (torch.bfloat16,):
    for (H, W, Cin, mid) in ((30, 40, 112, 448), (30, 40, 112, 672), (30, 40, 136, 816), (15, 20, 232, 1392)):
        M = B * H * W
        x = ops.Act(torch.randn(M, Cin, device=DEV).to(dt), 0, Cin)
        wpw = (torch.randn(mid, Cin) / math.sqrt(Cin)).to(dt)
        wdw = (torch.randn(9, mid) * 0.3).to(dt).to(DEV)
        s1 = torch.ones(mid, device=DEV); t1 = torch.zeros(mid, device=DEV)
        midb = ops.new_act(M, mid, dt, DEV); out = ops.new_act(M, mid, dt, DEV)
        wimg = ops.pack_mbconv_pw(wpw.float(), dt).to(DEV)
        wpwd = wpw.to(DEV)
        ns = ops.dwconv3x3_strips(B, H, W, mid, 1, ops.DT[dt])
        part = torch.empty(B * max(ns, ops.mbconv_plan(B, H, W, Cin, mid)[0]) * mid, device=DEV)

        def separate():
            ops.conv2d(x, wpwd, s1, t1, midb, B, H, W, 1, 1, 1, 0, 0, H, W, hip.ACT_SILU)
            ops.dwconv3x3_sum(midb, wdw, s1, t1, out, part, B, H, W, 1, 1, 1, H, W, hip.ACT_SILU)

        def fused():
            ops.mbconv_expand_dw(x, wimg, s1, t1, wdw, s1, t1, out, part, B, H, W)
        r = []
        for name, fn in (("expand GEMM + depthwise kernel", separate), ("fused", fused)):
            r.append((name, graph_time_us(fn, calls=8, replays=5), graph_time_us_concurrent(fn, streams, calls=8, replays=5)))
        print(f"{H}x{W} Cin {Cin} mid {mid}: " + " | ".join(f"{n}: {a:6.1f} us alone, {c:6.1f} us with 4 side by side" for n, a, c in r), flush=True)
