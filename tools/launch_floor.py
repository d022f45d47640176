#!/usr/bin/env python3
"""What does one more (tiny) kernel cost inside a replayed HIP graph on this box?  Captures N
dependent launches of a trivial C-ABI kernel (row copy of 64 rows) and times the replay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops
hip.load()
dev = "cuda:0"
a = ops.new_act(64, 64, torch.bfloat16, dev, zero=True)
b = ops.new_act(64, 64, torch.bfloat16, dev, zero=True)
big_a = ops.new_act(9600, 816, torch.bfloat16, dev, zero=True)
big_b = ops.new_act(9600, 816, torch.bfloat16, dev, zero=True)
for name, (x, y, rows) in {"tiny (8 KB)": (a, b, 64), "15.7 MB copy": (big_a, big_b, 9600)}.items():
    for N in (50, 200, 800):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                ops.copy_rows(x, y, rows); ops.copy_rows(y, x, rows)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(N // 2):
                ops.copy_rows(x, y, rows); ops.copy_rows(y, x, rows)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 20
        for _ in range(R): g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / R
        # eager back-to-back for comparison
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(N // 2):
            ops.copy_rows(x, y, rows); ops.copy_rows(y, x, rows)
        torch.cuda.synchronize()
        de = time.perf_counter() - t0
        print(f"{name:14s} N={N:4d}: graph replay {dt * 1e6 / N:6.2f} us/kernel   eager {de * 1e6 / N:6.2f} us/kernel", flush=True)
