#!/usr/bin/env python3
"""The adaptive-bins head at the bench shape (batch 8, 240x320 half-resolution map): separate kernels (3x3 conv -> ram -> fused
conv_out + softmax) against the one-kernel head (csrc/head_fused.hip) with its exactness islands on / off; back-to-back inside a
replayed HIP graph, alone and with 4 copies side by side (the throughput mode of bench.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from cfpnet_amd.engine import concurrent_streams
from _gtime import graph_time_us, graph_time_us_concurrent
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 240, 320
M = B * H * W
FL = 2.0 * M * 128 * (9 * 128 + 256)
for dt in (torch.bfloat16, torch.float16):
    x = ops.Act((torch.randn(M, 128, device=DEV)).to(dt), 0, 128)
    w3 = (torch.randn(128, 9 * 128, device=DEV) * 0.03).to(dt)
    wo = torch.randn(256, 128) * 0.3
    wo16 = wo.to(dt).to(DEV)
    sc, sh = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    bo = torch.zeros(256, device=DEV)
    cen = torch.sort(torch.rand(B, 256, device=DEV) * 10, dim=1)[0].contiguous()
    ram = ops.new_act(M, 128, dt, DEV)
    prob = torch.empty(B, 256, H * W, dtype=dt, device=DEV)
    pred = torch.empty(M, device=DEV)

    def unfused():
        ops.conv2d(x, w3, sc, sh, ram, B, H, W, 3, 3, 1, 1, 1, H, W)
        ops.bin_head_fused(ram, wo16, bo, cen, prob, pred, B, H * W)
    rows = [("separate kernels (conv3x3 + bin_head_fused)", unfused)]
    for hl in ((False, False), (True, False), (True, True)):
        wp = ops.permute_wout(wo, dt, hilo=hl[0]).to(DEV)
        rows.append((f"one kernel, Wout hi+lo={hl[0]}, ram hi+lo={hl[1]}",
                     lambda wp=wp, hl=hl: ops.depth_head_fused(x, w3, sc, sh, wp, bo, cen, prob, pred, B, H, W, ram_hilo=hl[1])))
    wp0 = ops.permute_wout(wo, dt, hilo=False).to(DEV)
    for probe, what in ((1, "probe: no fetch (zero-record descriptors)"), (2, "probe: GEMM1 only"), (3, "probe: GEMM1 only, no fetch"), (4, "probe: GEMM1 + GEMM2"),
                        (5, "probe: GEMM1 + GEMM2, no fetch")):
        rows.append((what, lambda probe=probe: ops.depth_head_fused(x, w3, sc, sh, wp0, bo, cen, prob, pred, B, H, W, ram_hilo=False, probe=probe)))
    rows.append(("no prob output", lambda: ops.depth_head_fused(x, w3, sc, sh, wp0, bo, cen, None, pred, B, H, W, ram_hilo=False)))
    streams = concurrent_streams(DEV, 4)
    for name, fn in rows:
        t = graph_time_us(fn, calls=6, replays=5)
        t4 = graph_time_us_concurrent(fn, streams, calls=6, replays=5)
        print(f"{str(dt):16s} {name:52s} alone {t:7.1f} us ({FL / t / 1e6:5.0f} TFLOP/s)   4 side by side {t4:7.1f} us per call")
