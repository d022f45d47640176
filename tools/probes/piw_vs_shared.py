#!/usr/bin/env python3
"""Upper bound of the gate-in-the-A-path idea (round 5, not built): the f16x3 project GEMMs with per-image weights (what se_gate_fold2 writes) against
the same GEMMs with shared weights, alone and four copies side by side, batch 8 and 1; plus se_gate_fold2's own time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops
from cfpnet_amd.engine import concurrent_streams
from _gtime import graph_time_us, graph_time_us_concurrent
hip.load()
DEV = "cuda:0"
for B, infl in ((8, 1), (8, 4), (1, 1)):
    streams = concurrent_streams(DEV, want=infl) if infl > 1 else None
    t = (lambda fn: graph_time_us_concurrent(fn, streams, calls=8, replays=3)) if streams else (lambda fn: graph_time_us(fn, calls=8, replays=4))
    ops.PLAN_IN_FLIGHT = infl > 1
    tot = [0.0, 0.0, 0.0]
    for (hw, C, Cout, R, n) in ((1200, 448, 112, 28, 4), (1200, 672, 136, 28, 1), (1200, 816, 136, 34, 6), (300, 816, 232, 34, 1), (300, 1392, 232, 58, 11)):
        M = B * hw
        x = ops.Act(torch.randn(M, C, device=DEV), 0, C)
        w = torch.randn(Cout, C, device=DEV) / C ** 0.5
        wx = ops.pack_w_x3(w)
        wb = torch.zeros(B, Cout, (C + 31) // 32 * 64, dtype=torch.float16, device=DEV)
        wb[:] = wx[None]
        out = ops.new_act(M, Cout, torch.float32, DEV)
        sc, sh = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
        _, sp = ops.conv2d_plan(M, Cout, C, hip.F32X3, hw, B, 1, 1)
        ws = torch.empty(max(sp, 1) * M * Cout, device=DEV) if sp > 1 and infl == 1 else None
        _, sp2 = ops.conv2d_plan(M, Cout, C, hip.F32X3, 0, B, 1, 1)
        ws2 = torch.empty(max(sp2, 1) * M * Cout, device=DEV) if sp2 > 1 else None
        piw = lambda: ops.conv2d(x, wb, sc, sh, out, B, 1, hw, 1, 1, 1, 0, 0, 1, hw, hip.ACT_NONE, None, ws, per_image_weights=True)
        shared = lambda: ops.conv2d(x, wx, sc, sh, out, 1, 1, M, 1, 1, 1, 0, 0, 1, M, hip.ACT_NONE, None, ws2)
        K = 90
        hpart = torch.randn(B * K * R, device=DEV)
        br, we_t, be = torch.randn(R, device=DEV), torch.randn(R, C, device=DEV), torch.randn(C, device=DEV)
        fold = lambda: ops.se_gate_fold2(hpart, K, 1.0 / hw, br, we_t, be, w, wb, B, Cout, C, R, x3=True)
        a, b_, c = t(piw), t(shared), t(fold)
        tot[0] += n * a; tot[1] += n * b_; tot[2] += n * c
        print(f"B={B} x{infl}: {M:5d} x {Cout:3d} x {C:4d}  per-image {a:6.1f} us   shared {b_:6.1f} us   fold2 {c:6.1f} us   (x{n})", flush=True)
    print(f"B={B} x{infl} per forward (24 blocks): per-image {tot[0]:.0f} us, shared {tot[1]:.0f} us, fold2 {tot[2]:.0f} us -> the idea saves at most {tot[0] - tot[1] + tot[2]:.0f} us minus a gate kernel", flush=True)
