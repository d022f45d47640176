#!/usr/bin/env python3
"""The latency-chain kernels of the forward at the bench shape (batch 8), back-to-back in a replayed graph: ToF encoder, the regressor
branch (channel sums of t, conv3x3_mean, bin_regressor), squeeze-excite gate fold."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops, spec, synthetic, weights
from cfpnet_amd.engine import Engine
from _gtime import graph_time_us
DEV = "cuda:0"
B = 8
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
eng = Engine(sd, layer_names=layers, dtype=torch.bfloat16)
dt = torch.bfloat16
R = B * 64 * 16
hist = torch.rand(R, device=DEV) * 3
outs = [ops.new_act(R, c, dt, DEV) for c in (32, 64, 128)]
pe = [eng.P[f"decoder.cross_atten{ex}.pe2"] for ex in (1, 2, 3)]
print(f"hist_encoder            {graph_time_us(lambda: ops.hist_encoder(hist, eng.P['hist.blob'], eng._hist_layout, outs, R, pe, 16)):7.1f} us")
H, W = 240, 320
t = ops.Act(torch.randn(B * H * W, 32, device=DEV).to(dt), 0, 32)
ns = 64
bsum = torch.empty(B * ns * 32, device=DEV); msum = torch.empty(B * 128, device=DEV)
edges = torch.empty(B, 257, device=DEV); cen = torch.empty(B * 256, device=DEV)
print(f"channel_sum(t)          {graph_time_us(lambda: ops.channel_sum(t, bsum, B, H * W, ns)):7.1f} us")
print(f"conv3x3_mean            {graph_time_us(lambda: ops.conv3x3_mean(bsum, ns, t, eng.P['decoder.conv0.w32'], eng.P['decoder.conv0.b32'], msum, B, H, W, 128)):7.1f} us")
h = "depth_head"
print(f"bin_regressor           {graph_time_us(lambda: ops.bin_regressor(msum, 1, 1.0 / (H * W), eng.P[h + '.w1x1'], eng.P[h + '.r0.w'], eng.P[h + '.r0.b'], eng.P[h + '.r2.w'], eng.P[h + '.r2.b'], eng.P[h + '.r4.w'], eng.P[h + '.r4.b'], 1e-3, 10.0, 0, edges, cen, B, 128, 256, 256)):7.1f} us")
b = [x for x in spec.ENC_BLOCKS if x.kind == 'ir' and x.mid == 816 and x.stride == 1][0]
q = "img_encoder." + b.prefix
nss = ops.dwconv3x3_strips(B, 30, 40, b.mid, 1, ops.DT[dt])
part = torch.rand(B * nss * b.mid, device=DEV)
wb = torch.empty(B, b.cout, b.mid, dtype=dt, device=DEV)
print(f"se_gate_fold (C=816)    {graph_time_us(lambda: ops.se_gate_fold(part, nss, 1.0 / 1200, eng.P[q + '.se.wr'], eng.P[q + '.se.br'], eng.P[q + '.se.we_t'], eng.P[q + '.se.be'], eng.P[q + '.pwl.w'], wb, B, b.cout, b.mid, b.se_rd)):7.1f} us  ({nss} strips)")

# LKPM MLP at the three fusion scales: LayerNorm + pwconv1 + pwconv2 (three launches, 4D-wide tensor through HBM) vs cfp_lkpm_tail
for (rows, D) in ((B * 120 * 160, 32), (B * 60 * 80, 64), (B * 30 * 40, 128)):
    t1 = ops.Act(torch.randn(rows, D, device=DEV).abs().to(dt), 0, D); xin = ops.Act(torch.randn(rows, D, device=DEV).to(dt), 0, D)
    t2 = ops.new_act(rows, D, dt, DEV); h4 = ops.new_act(rows, 4 * D, dt, DEV); o = ops.new_act(rows, D, dt, DEV)
    w1 = (torch.randn(4 * D, D, device=DEV) * 0.1).to(dt); w2 = (torch.randn(D, 4 * D, device=DEV) * 0.05).to(dt)
    b1 = torch.zeros(4 * D, device=DEV); b2 = torch.zeros(D, device=DEV); g1 = torch.ones(D, device=DEV); one4 = torch.ones(4 * D, device=DEV)

    def sep():
        ops.layernorm(t1, g1, b2, 1e-6, t2, rows)
        ops.linear(t2, w1, one4, b1, h4, rows, hip.ACT_GELU)
        ops.linear(h4, w2, g1, b2, o, rows, hip.ACT_NONE, xin)
    print(f"LKPM MLP rows {rows} D {D}: separate {graph_time_us(sep):7.1f} us | fused {graph_time_us(lambda: ops.lkpm_tail(t1, xin, o, w1, b1, w2, b2, g1, b2, rows)):7.1f} us")
