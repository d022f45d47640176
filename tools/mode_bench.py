#!/usr/bin/env python3
"""Batch-8 480x640 forward (prob output included) in the three storage modes, same protocol as bench.py:
HIP-graph replay, single graph (latency) and the best in-flight mode (throughput)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine

for kv in filter(None, os.environ.get("CFP_DEBUG_SET", "").split(",")):      # e.g. CFP_DEBUG_SET=27=1: debug knobs for A/B runs
    k, v = kv.split("=")
    from cfpnet_amd import hip as _hip
    _hip.load().cfp_debug_set(int(k), int(v))
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
inp = synthetic.to_device(synthetic.make_inputs(B), "cuda:0")
_ALL = {"f32": torch.float32, "x3": "x3", "f16": torch.float16, "bf16": torch.bfloat16}
_sel = os.environ.get("MODE_BENCH_DTYPES")    # e.g. "f32" or "f32,f16"
for dt in ((torch.bfloat16,) if os.environ.get('MODE_BENCH_BF16_ONLY') else tuple(_ALL[k] for k in _sel.split(",")) if _sel else tuple(_ALL.values())):
    eng = Engine(sd, layer_names=layers, dtype=torch.float32, x3=True) if dt == "x3" else Engine(sd, layer_names=layers, dtype=dt)
    if os.environ.get("MODE_BENCH_CHECK"):      # relative L1 of the depth map against the float32 engine's
        e32 = Engine(sd, layer_names=layers, dtype=torch.float32)
        p32 = e32.forward(inp)[1].float()
        pm = eng.forward(inp)[1].float()
        rel = ((pm - p32).abs().flatten(1).sum(1) / p32.abs().flatten(1).sum(1))
        print(json.dumps({"dtype": str(dt), "rel_l1_vs_f32_engine_per_image": [float(v) for v in rel]}))
        del e32
    mx = int(os.environ.get("MODE_BENCH_MAX_INFLIGHT", "4"))
    (kind, n), times = eng.capture_best(inp, reps=12, candidates=(("lanes", 1),) + tuple(("inflight", k) for k in range(2, mx + 1)))
    best = min(times.values())
    print(json.dumps({"dtype": str(dt), "batch": B, "choice": f"{kind}:{n}", "ms_per_step": times,
                      "maps_per_s_best": B / best * 1e3, "maps_per_s_single_graph": B / times["lanes:1"] * 1e3}))
    del eng
    torch.cuda.empty_cache()
