#!/usr/bin/env python3
"""Golden vectors for the evaluation metrics: runs THE REFERENCE's `compute_errors`
(`/root/reference/src/utils/metrics.py`) on seeded synthetic prediction / ground-truth pairs pushed through the two
protocols of `evaluate_all.py:38-41,80-84` and `train.py:187-199`.  Build container only.

    python oracle/gen_golden_metrics.py        -> tests/golden/eval_metrics.json
"""
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cfpnet_amd import synthetic  # noqa: E402
from oracle import metrics_oracle as MO  # noqa: E402  (protocol glue only; the metric itself comes from the reference)

CASES = [dict(name="eval480", H=480, W=640, Hp=240, Wp=320, seed=51, holes=0.0, noise=0.05, lo=1e-3, hi=10.0),
         dict(name="eval480_holes", H=480, W=640, Hp=240, Wp=320, seed=52, holes=0.3, noise=0.15, lo=1e-3, hi=10.0),
         dict(name="train416", H=416, W=544, Hp=208, Wp=272, seed=53, holes=0.2, noise=0.3, lo=1e-3, hi=10.0),
         dict(name="tight_range", H=480, W=640, Hp=240, Wp=320, seed=54, holes=0.1, noise=0.5, lo=1.0, hi=3.0),
         dict(name="same_size", H=240, W=320, Hp=240, Wp=320, seed=55, holes=0.0, noise=0.1, lo=1e-3, hi=10.0)]


def main():
    spec = importlib.util.spec_from_file_location("ref_metrics", "/root/reference/src/utils/metrics.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = []
    for c in CASES:
        gt, pred = synthetic.make_eval_pair(c["H"], c["W"], c["Hp"], c["Wp"], c["seed"], c["holes"], c["noise"])
        rec = dict(c)
        g, p = MO.protocol_evaluate_all(pred, gt, c["lo"], c["hi"])
        rec["evaluate_all"] = {k: float(v) for k, v in ref.compute_errors(g, p).items()}
        g, p = MO.protocol_validate(pred, gt, c["lo"], c["hi"])
        rec["validate"] = {k: float(v) for k, v in ref.compute_errors(g, p).items()}
        rec["n_valid"] = int(g.size)
        out.append(rec)
        print(c["name"], rec["n_valid"], {k: round(v, 5) for k, v in rec["evaluate_all"].items()})
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "eval_metrics.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
