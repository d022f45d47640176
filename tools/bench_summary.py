#!/usr/bin/env python3
"""Headline fields of a bench.py line: python tools/bench_summary.py gpurun_out/<tag>/bench_b8.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("maps/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "rel_l1", d.get("rel_l1"))
for k in ("f16", "f32"):
    if k in d:
        print(k, round(d[k]["value"], 1), d[k].get("rel_l1"))
print("latency", json.dumps(d.get("latency")))
if "training" in d:
    print("training", d["training"].get("value"), d["training"].get("ms_per_step"))
print("roofline", d["roofline"]["frac"], "dw in graph", d["dw3x3"].get("frac_of_measured_copy_in_graph"))
