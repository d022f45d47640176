#!/usr/bin/env python3
"""Graph-timed launches of the two data-path kernels either side of the model:
the ToF zone-histogram simulation and the evaluation metrics, at the headline geometry (B x 480x640)."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import metrics, synthetic, tof
from _gtime import graph_time_us
import types

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle on one image")
a = ap.parse_args()
B, H, W = a.batch, 480, 640
cfg = types.SimpleNamespace(mode="online_eval", train_zone_num=8, train_zone_random_offset=0, simu_max_distance=4.0,
                            zone_sample_num=16, sample_uniform=True)
deps = np.stack([synthetic.make_depth(H, W, seed=900 + i, holes=0.1 * (i % 3)) for i in range(B)])
d = torch.from_numpy(deps).cuda()
sim = tof.TofSimulator(cfg, "cuda:0")
out = sim.simulate(d)
t_tof = graph_time_us(lambda: sim.simulate(d, out=out), calls=8, replays=6)
zone_bytes = B * 64 * 56 * 56 * 4
pred = torch.from_numpy(np.stack([synthetic.make_eval_pair(H, W, 240, 320, 700 + i, 0.1, 0.1)[1] for i in range(B)])).cuda()
rows = torch.empty(B, 10, dtype=torch.float64, device="cuda:0")
metrics.eval_metrics(pred, d, 1e-3, 10.0, out=rows)
t_met = graph_time_us(lambda: metrics.eval_metrics(pred, d, 1e-3, 10.0, out=rows), calls=8, replays=6)
met_bytes = B * (H * W + 240 * 320) * 4
res = dict(batch=B, tof_us=t_tof, tof_GBps=zone_bytes / t_tof * 1e-3, metrics_us=t_met, metrics_GBps=met_bytes / t_met * 1e-3)
if a.cpu:
    from oracle import tof_oracle, metrics_oracle
    t0 = time.perf_counter(); tof_oracle.get_hist(deps[0]); res["tof_oracle_ms_per_image"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    g, p = metrics_oracle.protocol_evaluate_all(pred[0].cpu().numpy(), deps[0], 1e-3, 10.0); metrics_oracle.compute_errors(g, p)
    res["metrics_oracle_ms_per_image"] = (time.perf_counter() - t0) * 1e3
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/data_path_bench.json", "w"))
