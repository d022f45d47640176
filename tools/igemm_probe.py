#!/usr/bin/env python3
"""Where a small implicit GEMM's K loop spends its time: the same launch with the operand DMA, the fragment reads + MFMAs, or both
switched off (cfp_debug_set key 16; outputs are garbage in those modes)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
DEV = "cuda:0"
lib = hip.load()
dt = torch.bfloat16
CASES = [(9600, 136, 816, 13), (9600, 136, 816, 4), (2400, 232, 1392, 13), (9600, 816, 136, 13), (38400, 64, 1152, 13), (9600, 136, 816, 14),
         (9600, 112, 448, 13), (9600, 112, 448, 15), (9600, 112, 448, 12), (9600, 136, 816, 15), (9600, 128, 512, 13), (9600, 128, 512, 15), (9600, 128, 512, 14), (9600, 128, 512, 1)]
for M, N, K, variant in CASES:
    x = ops.Act(torch.randn(M, K, device=DEV).to(dt), 0, K)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dt)
    out = ops.new_act(M, N, dt, DEV)
    fn = lambda: ops.conv2d(x, w, None, None, out, 1, 1, M, 1, 1, 1, 0, 0, 1, M)
    lib.cfp_debug_set(0, variant)
    row = []
    for probe in (0, 1, 2, 3):
        lib.cfp_debug_set(16, probe)
        fn(); torch.cuda.synchronize()
        row.append(min(graph_time_us(fn, calls=16, replays=5) for _ in range(2)))
    lib.cfp_debug_set(16, 0); lib.cfp_debug_set(0, -1)
    nk = (K + 63) // 64
    print(f"{M:6d} x {N:4d} x {K:5d}  tile v{variant:<2d} {nk:3d} K-steps: full {row[0]:6.1f} us   no DMA {row[1]:6.1f}   no reads/MFMA {row[2]:6.1f}   neither {row[3]:6.1f}")
