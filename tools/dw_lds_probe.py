#!/usr/bin/env python3
"""Launch the depthwise 3x3 MFMA kernel on one encoder shape, normally or with its compute phase skipped (act = 99: staging and
copy-out only), for a `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE` pass: which phase the bank conflicts come from.
    python tools/dw_lds_probe.py full|nocompute [H W C]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops
mode = sys.argv[1]
H, W, C = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (30, 40, 816)
B, DEV = 8, "cuda:0"
x = ops.Act(torch.randn(B * H * W, C, device=DEV).to(torch.bfloat16), 0, C)
out = ops.new_act(B * H * W, C, torch.bfloat16, DEV)
w = torch.randn(9, C, device=DEV).to(torch.bfloat16); sc = torch.ones(C, device=DEV); sh = torch.zeros(C, device=DEV)
part = torch.zeros(B * 64 * C, device=DEV)
for _ in range(20):
    ops.dwconv3x3_sum(x, w, sc, sh, out, part, B, H, W, 1, 1, 1, H, W, 99 if mode == "nocompute" else hip.ACT_SILU)
torch.cuda.synchronize()
