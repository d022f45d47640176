import os, sys, collections
import torch
sys.path.insert(0, '/root/repo')
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.engine import Engine
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
eng = Engine(sd, layer_names=layers, dtype=torch.float32, x3=True)
inp = synthetic.to_device(synthetic.make_inputs(1), "cuda:0")
for _ in range(2):
    eng.forward(inp)
torch.cuda.synchronize()
real = hip.call
names = []
def rec(name, *a):
    names.append(name); real(name, *a)
hip.call = rec
eng.forward(inp)
torch.cuda.synchronize()
hip.call = real
c = collections.Counter(names)
print(len(names), "C-ABI calls")
for k, v in c.most_common():
    print(f"{v:4d} {k}")
