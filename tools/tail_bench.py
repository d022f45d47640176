#!/usr/bin/env python3
"""cfp_loftr_tail at the three fusion scales of the benched batch (graph-timed, back-to-back launches)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
DEV = "cuda:0"
dt = torch.bfloat16
g = torch.Generator().manual_seed(1)
for D, heads, NB, Hq, Wq, qt in [(128, 8, 8, 30, 40, 4), (128, 4, 8, 30, 40, 4), (64, 8, 8, 60, 80, 8), (64, 4, 8, 60, 80, 8), (32, 8, 8, 120, 160, 15), (32, 4, 8, 120, 160, 15)]:
    rows = NB * Hq * Wq
    d = D // heads
    G = NB * ((Hq + qt - 1) // qt) * ((Wq + qt - 1) // qt)
    x = ops.new_act(rows, D, dt, DEV); x.buf.copy_(torch.randn(rows, D, generator=g).to(dt))
    out = ops.new_act(rows, D, dt, DEV)
    kv = torch.randn(G * heads, d, d, generator=g).to(DEV); ks = torch.rand(G * heads, d, generator=g).to(DEV) + 0.5
    mk = lambda n, k: (torch.randn(n, k, generator=g) / k ** 0.5).to(dt).to(DEV)
    wq, wm, w0, w2 = mk(D, D), mk(D, D), mk(2 * D, 2 * D), mk(D, 2 * D)
    ln = lambda: (torch.ones(D, device=DEV), torch.zeros(D, device=DEV), 1e-5)
    ln1, ln2 = ln(), ln()
    fn = lambda: ops.loftr_tail(None, kv, ks, x, out, wq, wm, w0, w2, ln1, ln2, NB, Hq, Wq, qt, qt, float(qt * qt), heads)
    fn(); torch.cuda.synchronize()
    t = min(graph_time_us(fn, calls=16, replays=5) for _ in range(2))
    print(f"D={D:3d} heads={heads} rows={rows:6d}: {t:6.1f} us per launch")
