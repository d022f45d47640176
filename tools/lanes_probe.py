#!/usr/bin/env python3
"""Batch 8 cut into 1 / 2 / 3 / 4 sub-batch graphs launched together (Engine.capture(lanes=n)), timed back to back WITHOUT a
synchronise between forwards: lanes:4 3.78-3.80 ms vs 4.01 ms for one graph.  Under the reference's latency protocol (each forward
bracketed by a device synchronise, bench.py::reference_latency_ms) the same split measures 6.5 ms: four graph launches from an idle
queue cost more host time than the overlap wins, so the latency figures of the bench line stay single-graph."""
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
inp = synthetic.to_device(synthetic.make_inputs(8), "cuda:0")
eng = Engine(sd, layer_names=layers, dtype=torch.bfloat16)
for tput in (False, True):
    eng.plan_mode(tput)
    (kind, n), times = eng.capture_best(inp, reps=30, candidates=(("lanes", 1), ("lanes", 2), ("lanes", 3), ("lanes", 4)), allow_inflight=False)
    print(json.dumps({"tput_plan": tput, "times": times}))
