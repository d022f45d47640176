import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
lib = hip.load()
B, HW, Cin = 8, 240 * 320, 128
x = ops.Act(torch.randn(B * HW, Cin, device="cuda"), 0, Cin)
wx = ops.pack_w_x3(torch.randn(256, Cin, device="cuda") * 0.1)
bias = torch.randn(256, device="cuda"); cen = torch.rand(B, 256, device="cuda")
prob = torch.empty(B, 256, HW, device="cuda"); pred = torch.empty(B, HW, device="cuda")
for rows in (64, 128):
    lib.cfp_debug_set(25, rows)
    fn = lambda: ops.bin_head_fused(x, wx, bias, cen, prob, pred, B, HW)
    print(rows, "rows per workgroup:", round(graph_time_us(fn, calls=8, replays=4), 1), "us with prob;",
          round(graph_time_us(lambda: ops.bin_head_fused(x, wx, bias, cen, None, pred, B, HW), calls=8, replays=4), 1), "us without")
