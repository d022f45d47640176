#!/usr/bin/env python3
"""Where an f16x3 implicit GEMM's K loop spends its time: the same launch with parts switched off (cfp_debug_set key 16; outputs are
garbage in those modes): 1 = no operand DMA after the prologue, 2 = no fragment reads / MFMAs, 3 = neither, 4 = no hi / lo split and one
MFMA per block, 8 = split but one MFMA per block."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
DEV = "cuda:0"
lib = hip.load()
# (B, H, W, Cin, Cout, k, variant)
CASES = [(8, 240, 320, 128, 128, 3, 26), (8, 240, 320, 128, 128, 3, 1), (8, 240, 320, 128, 256, 1, 26), (8, 120, 160, 40, 160, 3, 14), (8, 30, 40, 136, 816, 1, 13),
         (8, 30, 40, 816, 136, 1, 13), (8, 120, 160, 64, 32, 1, 16), (8, 60, 80, 128, 64, 3, 13), (8, 60, 80, 128, 64, 3, 14), (8, 15, 20, 1392, 232, 1, 19), (8, 240, 320, 80, 32, 3, 16)]
for B, H, W, Cin, Cout, k, variant in CASES:
    M, K = B * H * W, k * k * Cin
    x = ops.Act(torch.randn(M, Cin, device=DEV), 0, Cin)
    w = ops.pack_w_x3((torch.randn(Cout, K, device=DEV) / math.sqrt(K)).contiguous())
    out = ops.new_act(M, Cout, torch.float32, DEV)
    fn = lambda: ops.conv2d(x, w, None, None, out, B, H, W, k, k, 1, k // 2, k // 2, H, W)
    lib.cfp_debug_set(0, 400 + variant)
    row = []
    for probe in (0, 1, 2, 3, 4, 8):
        lib.cfp_debug_set(16, probe)
        fn(); torch.cuda.synchronize()
        row.append(min(graph_time_us(fn, calls=12, replays=4) for _ in range(2)))
    lib.cfp_debug_set(16, 0)
    lib.cfp_debug_set(28, 1)      # the plain K loop (probes above force it too; "full" above is the fragment-pipelined loop)
    fn(); torch.cuda.synchronize()
    plain = min(graph_time_us(fn, calls=12, replays=4) for _ in range(2))
    lib.cfp_debug_set(28, 0); lib.cfp_debug_set(0, -1)
    print(f"{M:6d} x {Cout:4d} x {K:5d}  tile v{variant:<2d} {(K + 31) // 32:3d} K-steps: full {row[0]:6.1f} us   no DMA {row[1]:6.1f}   no reads/MFMA {row[2]:6.1f}   "
          f"neither {row[3]:6.1f}   no split, 1 MFMA {row[4]:6.1f}   split, 1 MFMA {row[5]:6.1f}   | plain loop, full {plain:6.1f}")
