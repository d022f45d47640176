#!/usr/bin/env python3
"""loftr_tail_x3 on config5's LSA geometry (2 x 160 x 240 tokens, 14 x 14 windows, D = 32, 8 heads), 20 runs: bit-stable?"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfpnet_amd import hip, ops
DEV = "cuda:0"
torch.manual_seed(0)
for (D, heads, NB, Hq, Wq, qth, qtw) in [(32, 8, 2, 160, 240, 14, 14), (32, 8, 2, 160, 240, 160, 240), (32, 8, 8, 120, 160, 7, 7), (64, 8, 2, 80, 120, 10, 10), (128, 8, 2, 40, 60, 7, 7), (32, 4, 2, 160, 240, 14, 14)]:
    d = D // heads
    rows = NB * Hq * Wq
    G = NB * (-(-Hq // qth)) * (-(-Wq // qtw))
    x = ops.Act(torch.randn(rows, 2 * D, device=DEV), 0, D)
    kv = torch.randn(G, heads, d, d, device=DEV) * 0.3
    ks = torch.rand(G, heads, d, device=DEV) + 0.5
    P = lambda n, k: ops.pack_w_x3((torch.randn(n, k, device=DEV) / math.sqrt(k)).contiguous())
    wq, wm, w0, w2 = P(D, D), P(D, D), P(2 * D, 2 * D), P(D, 2 * D)
    ln = lambda: (torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV))
    ln1, ln2 = ln(), ln()
    for own_q in (True, False):
        qa = None if own_q else ops.Act(torch.randn(rows, 3 * D, device=DEV), 0, D)
        ref, stable, nbad = None, True, 0
        for i in range(20):
            out = ops.new_act(rows, D, torch.float32, DEV)
            out.buf.fill_(float("nan"))
            ops.loftr_tail(qa, kv, ks, x, out, wq if own_q else None, wm, w0, w2, ln1, ln2, NB, Hq, Wq, qth, qtw, float(qth * qtw), heads)
            torch.cuda.synchronize()
            if ref is None: ref = out.buf.clone()
            elif not torch.equal(ref, out.buf):
                stable = False; nbad = int((ref != out.buf).any(1).sum())
        print(f"D={D} heads={heads} rows={rows} window {qth}x{qtw} own_q={own_q}: stable {stable} (rows differing in the last mismatch: {nbad}), finite {bool(torch.isfinite(ref).all())}")
