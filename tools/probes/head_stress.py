#!/usr/bin/env python3
"""Is the fused head kernel itself intermittently wrong?  The same launch repeated thousands of times (alone, and on 3 streams at once,
optionally with another kernel type interleaved), every output compared bit for bit with the first one."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfpnet_amd import hip, ops
from cfpnet_amd.engine import concurrent_streams
DEV = "cuda:0"
B, H, W = 2, 128, 160
M = B * H * W
dt = torch.float16
g = torch.Generator().manual_seed(0)
x = ops.Act((torch.randn(M, 128, generator=g)).to(dt).to(DEV), 0, 128)
w3 = (torch.randn(128, 9 * 128, generator=g) * 0.03).to(dt).to(DEV)
wo = torch.randn(256, 128, generator=g) * 0.3
wp = ops.permute_wout(wo, dt, hilo=False).to(DEV)
sc, sh, bo = torch.ones(128, device=DEV), torch.zeros(128, device=DEV), torch.zeros(256, device=DEV)
cen = torch.sort(torch.rand(B, 256, generator=g) * 10, dim=1)[0].contiguous().to(DEV)


xs = [x, ops.Act((torch.randn(M, 128, generator=g) * 1.7).to(dt).to(DEV), 0, 128)]


def run(prob, pred, which=0):
    ops.depth_head_fused(xs[which], w3, sc, sh, wp, bo, cen, prob, pred, B, H, W, ram_hilo=False)


prob0 = torch.empty(B, 256, H * W, dtype=dt, device=DEV); pred0 = torch.empty(M, device=DEV)
run(prob0, pred0)
torch.cuda.synchronize()
N = int(os.environ.get("PROBE_N", "3000"))
bad = 0
prob = torch.empty_like(prob0); pred = torch.empty_like(pred0)
for it in range(N):
    pred.zero_()
    run(prob, pred)
    if it % 50 == 49 or True:
        if not torch.equal(pred, pred0):
            bad += 1
            d = torch.nonzero(pred != pred0).flatten()
            if bad <= 5: print(f"alone: launch {it}: {d.numel()} wrong pixels, m {d[:3].tolist()}..{d[-3:].tolist()}", flush=True)
print(f"alone: {bad} wrong launches of {N}", flush=True)
streams = concurrent_streams(DEV, 3)
outs = [(torch.empty_like(prob0), torch.empty_like(pred0)) for _ in streams]
a = ops.Act(torch.randn(40960, 64, device=DEV).to(dt), 0, 64); wq = (torch.randn(192, 64, device=DEV) * 0.1).to(dt); oq = ops.new_act(40960, 192, dt, DEV)
bad = 0
for it in range(N // 3):
    for st, (pb, pd) in zip(streams, outs):
        with torch.cuda.stream(st):
            pd.zero_()
            ops.linear(a, wq, None, None, oq, 40960)          # another LDS-DMA kernel type in between
            run(pb, pd)
    torch.cuda.synchronize()
    for si, (pb, pd) in enumerate(outs):
        if not torch.equal(pd, pred0):
            bad += 1
            d = torch.nonzero(pd != pred0).flatten()
            if bad <= 5: print(f"3 streams: round {it} stream {si}: {d.numel()} wrong pixels, m {d[:3].tolist()}..{d[-3:].tolist()}", flush=True)
print(f"3 streams: {bad} wrong launches of {N // 3 * 3}", flush=True)
# alternating inputs: leftovers of the previous launch (LDS, caches) are now WRONG data for the current one
refs = []
for wch in (0, 1):
    run(prob0, pred0, wch); torch.cuda.synchronize(); refs.append(pred0.clone())
bad = 0
for it in range(N):
    for st, (pb, pd) in zip(streams, outs):
        with torch.cuda.stream(st):
            ops.linear(a, wq, None, None, oq, 40960)
            run(pb, pd, it & 1)
    torch.cuda.synchronize()
    for si, (pb, pd) in enumerate(outs):
        if not torch.equal(pd, refs[it & 1]):
            bad += 1
            d = torch.nonzero(pd != refs[it & 1]).flatten()
            if bad <= 5: print(f"alternating inputs: round {it} stream {si}: {d.numel()} wrong pixels, m {d[:3].tolist()}..{d[-3:].tolist()}", flush=True)
print(f"alternating inputs, 3 streams: {bad} wrong launches of {N * 3}", flush=True)
