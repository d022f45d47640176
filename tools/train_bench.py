#!/usr/bin/env python3
"""Step time of the HIP training step on a fixed device-resident batch (BASELINE.json configs[2..3] shape: 416x544 crops,
6x6 zones of 64 px, per-GPU batch 16): forward + SILog + backward + AdamW, float32."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.trainer import Trainer

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--dtype", default="f32", choices=("f32", "bf16", "f16")); ap.add_argument("--graph", action="store_true"); ap.add_argument("--beside", action="store_true", help="parameter-gradient kernels as graphs of their own on a second stream (Trainer.capture(wgrad_beside=True))"); ap.add_argument("--debug", default="", help="cfp_debug_set switches, e.g. 20=2048,21=4096")
a = ap.parse_args()
if a.debug:
    from cfpnet_amd import hip
    for kv in a.debug.split(","):
        k, v = kv.split("="); hip.load().cfp_debug_set(int(k), int(v))
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
H, W = 416, 544
inp = synthetic.to_device(synthetic.make_inputs(a.batch, H, W, 6, 64, seed=5, drop_hist=0.34), "cuda:0")
target = torch.from_numpy(np.stack([synthetic.make_depth(H, W, seed=50 + i, holes=0.1) for i in range(a.batch)]))[:, None].cuda()
DT = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}[a.dtype]
tr = Trainer(sd, layers, lr=3e-4, total_steps=100, dtype=DT)
if a.graph:
    tr.capture(inp, target, wgrad_beside=a.beside)
for _ in range(2):
    tr.step(inp, target)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    loss, _, _ = tr.step(inp, target)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
res = dict(batch=a.batch, ms_per_step=dt * 1e3, samples_per_s=a.batch / dt, loss=float(loss), peak_mem_GB=torch.cuda.max_memory_allocated() / 2**30, dtype=a.dtype, graph=a.graph)
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True); json.dump(res, open("gpurun_out/train_bench.json", "w"))
