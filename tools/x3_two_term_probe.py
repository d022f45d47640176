#!/usr/bin/env python3
"""VERDICT r4 task 1(b): would a TWO-term split product (A_hi W_hi + A_lo W_hi: weights held as ONE IEEE half -- W tile 2 B / element in
LDS, one MFMA fewer) stay inside the 1e-3 every-image gate?  Measured with the three-term kernels and the lo halves of every STATIC
weight zeroed at pack time (CFP_X3_DIAG_TWO_TERM=1; the 24 squeeze-excite-folded project weights stay three-term, so this is a LOWER
bound of the two-term error): relative L1 of the depth map against the float32 CPU oracle, every image of two batches, four weight
families -- the unchanged protocol of tests/test_forward_gpu.py::test_f32x3_meets_the_gate_on_every_image_of_every_weight_family.

    python tools/x3_two_term_probe.py            # prints one JSON line per (mode, family)
"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine
from oracle import cfpnet_oracle as O           # checker only
from oracle.calibrate import calibrate_bn

layers = spec.COMBINE1_LAYERS
torch.set_num_threads(max(torch.get_num_threads(), 8))
B = int(os.environ.get("PROBE_BATCH", "4"))
fams = {}
for fam in ("uniform", "kaiming", "kaiming_peaked", "trained"):
    if fam == "trained":
        sd = weights.trained_like_state_dict(layers, steps=300); sd.pop("__loss__", None)
    else:
        sd = weights.make_torch_state_dict(spec.model_manifest(layers), family=fam)
        if fam != "uniform":
            sd = calibrate_bn(sd, layers)
    inp = synthetic.make_inputs(B, 480, 640, 8, 56, seed=4242, drop_hist=0.34)
    p0 = O.forward(sd, inp, layer_names=layers)[1].numpy()
    dinp = synthetic.to_device(inp, "cuda:0")
    for mode in ("three_term", "two_term"):
        os.environ["CFP_X3_DIAG_TWO_TERM"] = "1" if mode == "two_term" else "0"
        eng = Engine(sd, layer_names=layers, dtype=torch.float32, x3=True)
        p1 = eng.forward(dinp)[1].cpu().numpy()
        per = [float(np.abs(p1[b] - p0[b]).sum() / np.abs(p0[b]).sum()) for b in range(B)]
        print(json.dumps({"family": fam, "mode": mode, "rel_l1_worst_image": max(per), "rel_l1_per_image": per, "gate": 1e-3, "gate_met": max(per) <= 1e-3}), flush=True)
        del eng
os.environ["CFP_X3_DIAG_TWO_TERM"] = "0"
