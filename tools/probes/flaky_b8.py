import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
inp = synthetic.to_device(synthetic.make_inputs(8, 480, 640, 8, 56, seed=synthetic.SEED), "cuda:0")
for dt in (torch.bfloat16, torch.float16):
    ref = None
    for it in range(6):
        eng = Engine(sd, layer_names=layers, dtype=dt)
        e1, p1, pr1 = eng.forward(inp)
        p1, e1 = p1.clone(), e1.clone()
        eng.capture(inp)
        e2, p2, pr2 = eng.replay()
        torch.cuda.synchronize()
        same = torch.equal(p1, p2) and torch.equal(e1, e2)
        if ref is None: ref = (p2.clone(), e2.clone())
        print(dt, it, "eager==replay", same, "edges equal", torch.equal(e1, e2), "same as first engine", torch.equal(ref[0], p2), torch.equal(ref[1], e2),
              "max |dp|", float((p1 - p2).abs().max()), "max |de|", float((e1 - e2).abs().max()), flush=True)
        del eng
