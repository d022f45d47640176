import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
calls = []
for (H, W, C, s, n) in [(60, 80, 224, 2, 1), (30, 40, 448, 1, 4), (30, 40, 672, 1, 1), (30, 40, 816, 1, 6), (30, 40, 816, 2, 1), (15, 20, 1392, 1, 11)]:
    Ho, Wo = -(-H // s), -(-W // s)
    pt = max((Ho - 1) * s + 3 - H, 0) // 2; pl = max((Wo - 1) * s + 3 - W, 0) // 2
    calls += [(8, H, W, C, s, pt, pl, Ho, Wo)] * n
from cfpnet_amd import hip; hip.load()
for dt in (torch.float32, torch.bfloat16):
    r = bench.dw3x3_at_hbm_scale(calls, dt, "cuda:0")
    print(dt, round(r["frac_of_measured_copy_rate"], 3), round(r["GBps"]), [(x["shape"], round(x["frac_of_measured_copy_rate"], 2)) for x in r["shapes"]])
    r = bench.dw3x3_in_graph(calls, dt, "cuda:0")
    print(dt, "in graph", round(r["frac_of_measured_copy_rate"], 3), round(r["us_per_step"], 1))
