#!/usr/bin/env python3
"""A/B of two builds of the library on the same forward: sha256 of (edges, pred, prob) of the default-mode engine at batch 1 and 8, and the
single-graph time.  `CFP_HIP_LIB=<other .so> python tools/probes/ab_forward_hash.py` for the other side; bit-identical kernels print the same hashes."""
import hashlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.engine import Engine

layers = spec.COMBINE1_LAYERS
x3 = os.environ.get("AB_DTYPE", "x3") == "x3"
dt = torch.float32 if x3 else {"bf16": torch.bfloat16, "f16": torch.float16}[os.environ["AB_DTYPE"]]
for fam in ("uniform",):      # (the kaiming families need calibrated BatchNorm statistics -- tests/helpers.calibrate_bn -- or they overflow)
    sd = weights.make_torch_state_dict(spec.model_manifest(layers), family=fam)
    eng = Engine(sd, layer_names=layers, dtype=dt, x3=x3)
    for B in (1, 8):
        inp = synthetic.to_device(synthetic.make_inputs(B, seed=3 + B), "cuda:0")
        outs = eng.forward(inp, return_prob=True)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for o in outs:
            if torch.is_tensor(o):
                h.update(o.detach().float().cpu().numpy().tobytes())
        eng.capture(inp, return_prob=True)
        for _ in range(5):
            eng.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            eng.replay()
        torch.cuda.synchronize()
        print(os.path.basename(hip.LIB_PATH), fam, "B", B, h.hexdigest()[:16], f"{(time.perf_counter() - t0) / 30 * 1e3:.3f} ms")
