#!/usr/bin/env python3
"""What does each kernel family cost in THROUGHPUT mode?  The forward is captured with one family's launches skipped
(results are garbage, timing is not) and replayed with 4 batches in flight; the difference to the full graph is the
family's marginal cost per step when everything else overlaps -- not its isolated duration."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.engine import Engine

layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
inputs = synthetic.to_device(synthetic.make_inputs(8), "cuda:0")
real = hip.call


def conv_gflop(args):
    B, H, W, Cin, Cout, KH, KW, st, pt, pl, Ho, Wo = args[9:21]
    return 2.0 * B * Ho * Wo * Cout * KH * KW * Cin / 1e9


def run(skip):
    def call(name, *a):
        if skip(name, a):
            return
        real(name, *a)
    hip.call = call
    try:
        e = Engine(sd, layer_names=layers, dtype=torch.bfloat16)
        e.capture(inputs, inflight=4)
    finally:
        hip.call = real
    for _ in range(12):
        e.replay_async()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        e.replay_async()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 40 * 1e3


is_conv = lambda n: n in ("cfp_conv2d_nhwc", "cfp_conv2d_nhwc_ex")
base = run(lambda n, a: False)
print(f"full graph: {base:.3f} ms/step")
cases = [("dw3x3", lambda n, a: n.startswith("cfp_dwconv3x3")), ("se_gate_fold", lambda n, a: n == "cfp_se_gate_fold"),
         ("loftr_tail", lambda n, a: n == "cfp_loftr_tail"), ("kv_reduce", lambda n, a: n == "cfp_attn_kv_reduce"),
         ("attn_apply", lambda n, a: n == "cfp_attn_apply"), ("resize", lambda n, a: n == "cfp_resize_bilinear"),
         ("bin_head_fused", lambda n, a: n == "cfp_bin_head_fused"), ("bin_regressor", lambda n, a: n == "cfp_bin_regressor"),
         ("dwlarge", lambda n, a: n.startswith("cfp_dwconv_large")), ("layernorm", lambda n, a: n == "cfp_layernorm"),
         ("conv < 0.5 GF", lambda n, a: is_conv(n) and conv_gflop(a) < 0.5), ("conv 0.5-2 GF", lambda n, a: is_conv(n) and 0.5 <= conv_gflop(a) < 2),
         ("conv 2-10 GF", lambda n, a: is_conv(n) and 2 <= conv_gflop(a) < 10), ("conv 10-60 GF", lambda n, a: is_conv(n) and 10 <= conv_gflop(a) < 60),
         ("conv >= 60 GF", lambda n, a: is_conv(n) and conv_gflop(a) >= 60), ("all convs", lambda n, a: is_conv(n))]
for tag, f in cases:
    t = run(f)
    print(f"without {tag:16s}: {t:.3f} ms/step   marginal cost {base - t:+.3f} ms ({(base - t) / base * 100:4.1f} %)", flush=True)
