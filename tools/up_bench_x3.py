#!/usr/bin/env python3
"""cfp_upsample_cat_conv3x3 in the f16x3 mode (two-source chunk kernel, round 5) against the pair it replaces (cfp_resize_bilinear + the f16x3
conv on the materialised concatenation) at the four decoder stages, batch 8 of 480x640; alone and with four copies side by side.
    python tools/up_bench_x3.py [--inflight 4]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from cfpnet_amd.engine import concurrent_streams
from _gtime import graph_time_us, graph_time_us_concurrent
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--inflight", type=int, default=1)
a = ap.parse_args()
hip.load()
DEV = "cuda:0"
B = a.batch
STREAMS = concurrent_streams(DEV, want=a.inflight) if a.inflight > 1 else None
def t(fn):
    return graph_time_us_concurrent(fn, STREAMS, calls=6, replays=3) if STREAMS else graph_time_us(fn, calls=6, replays=4)
for i, (Hs, Ws, Cup, Csk, Cout) in enumerate([(15, 20, 256, 136, 256), (30, 40, 256, 56, 128), (60, 80, 128, 40, 64), (120, 160, 64, 16, 32)], start=1):
    H, W = 2 * Hs, 2 * Ws
    low = ops.Act(torch.randn(B * Hs * Ws, Cup, device=DEV), 0, Cup)
    cat = ops.new_act(B * H * W, Cup + Csk, torch.float32, DEV); cat.buf.normal_()
    w = torch.randn(Cout, 3, 3, Cup + Csk, device=DEV) / (3 * (Cup + Csk) ** 0.5)
    wcat = ops.pack_w_x3_cat(w, Cup)
    wx = ops.pack_w_x3(w.reshape(Cout, -1).contiguous())
    sc, sh = torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV)
    out = ops.new_act(B * H * W, Cout, torch.float32, DEV)
    fused = lambda: ops.upsample_cat_conv3x3(low, Hs, Ws, cat.slice(Cup, Csk), wcat, sc, sh, out, B, H, W, hip.ACT_LRELU, x3=True)
    rs = lambda: ops.resize_bilinear(low, Hs, Ws, (0, 0, Hs, Ws), cat.slice(0, Cup), H, W, (0, 0, H, W), B)
    cv = lambda: ops.conv2d(cat, wx, sc, sh, out, B, H, W, 3, 3, 1, 1, 1, H, W, hip.ACT_LRELU)
    if STREAMS:
        ops.PLAN_IN_FLIGHT = True
    tf, tr, tc = t(fused), t(rs), t(cv)
    print(f"up{i}: {B}x{H}x{W} ({Cup}+{Csk}) -> {Cout}: fused {tf:7.1f} us   resize {tr:6.1f} + conv {tc:7.1f} = {tr + tc:7.1f} us", flush=True)
