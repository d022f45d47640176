#!/usr/bin/env python3
"""Workload for the FETCH_SIZE / WRITE_SIZE reconciliation of the depthwise 3x3 kernel: the encoder's largest
stride-1 depthwise shape and a plain row copy of the SAME tensor, 24 launches each on rotating buffers (so neither just hits in L2).
Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and feed both CSVs to tools/pmc_traffic.py: the copy's
counters calibrate the FETCH factor (its algorithmic bytes are known exactly), the depthwise kernel's are then read with that factor."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops
DEV = "cuda:0"
B, H, W, C, s = 8, 30, 40, 816, 1
NB = 8
xs = [ops.Act(torch.randn(B * H * W, C, device=DEV).to(torch.bfloat16), 0, C) for _ in range(NB)]
outs = [ops.new_act(B * H * W, C, torch.bfloat16, DEV) for _ in range(NB)]
w = torch.randn(9, C, device=DEV).to(torch.bfloat16); sc = torch.ones(C, device=DEV); sh = torch.zeros(C, device=DEV)
ns = ops.dwconv3x3_strips(B, H, W, C, s, hip.BF16)
part = torch.empty(B * ns * C, device=DEV)
for i in range(24):
    ops.dwconv3x3_sum(xs[i % NB], w, sc, sh, outs[i % NB], part, B, H, W, s, 1, 1, H, W, hip.ACT_SILU)
for i in range(24):
    ops.copy_rows(xs[i % NB], outs[(i + 3) % NB], B * H * W)
torch.cuda.synchronize()
print(f"algorithmic bytes per launch: copy {2 * B * H * W * C * 2} (read {B * H * W * C * 2} + write {B * H * W * C * 2}); "
      f"dw3x3 {2 * B * H * W * C * 2 + 9 * C * 2 + B * ns * C * 4} (same tensors + weights + channel-sum partials)")
