#!/usr/bin/env python3
"""Every conv / linear call of the f32x3 forward, re-issued 30 times on its recorded arguments: is the output bit-stable, and equal to the
plain (not fragment-pipelined) K loop's?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.engine import Engine
lib = hip.load()
base = (640, 960) if "--config5" in sys.argv else (480, 640)
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers, base_resolution=base))
inp = synthetic.to_device(synthetic.make_inputs(2, base[0], base[1], 16 if base[0] == 640 else 8, 40 if base[0] == 640 else 56, seed=31, drop_hist=0.2, image_hw=base), "cuda:0")
eng = Engine(sd, layer_names=layers, base_resolution=base)
eng.forward(inp); torch.cuda.synchronize()
calls, real = [], hip.call
def rec(name, *a):
    if name == "cfp_conv2d_nhwc_ex":
        calls.append(a)
    real(name, *a)
hip.call = rec
eng.forward(inp); torch.cuda.synchronize()
hip.call = real
import ctypes
seen = set()
for a in calls:
    B, H, W, Cin, Cout, KH, KW, st, pt, pl, Ho, Wo = a[9:21]
    key = (B, H, W, Cin, Cout, KH, st, bool(a[23]), a[26], bool(a[5]))
    if key in seen:
        continue
    seen.add(key)
    M, out_ld = B * Ho * Wo, a[8]
    n = M * out_ld
    view = (ctypes.c_float * 1).from_address  # noqa (unused)
    out_t = torch.empty(0)
    def out_tensor():
        # the recorded output pointer belongs to an engine buffer that is still alive: wrap it
        return torch.frombuffer((ctypes.c_char * 0).from_address(0), dtype=torch.float32) if False else None
    # run into a scratch output instead (same pitch), so nothing else is disturbed
    scratch = torch.empty(M, out_ld, device="cuda:0")
    args = list(a); args[7] = scratch.data_ptr()
    if a[5]:   # residual present: keep it (read only)
        pass
    outs = []
    for plain in (0, 1):
        lib.cfp_debug_set(28, plain)
        ref = None; stable = True
        for i in range(30):
            scratch.fill_(float("nan"))
            real("cfp_conv2d_nhwc_ex", *args[:-1], hip.current_stream())
            torch.cuda.synchronize()
            cur = scratch[:, :Cout].clone()
            if ref is None: ref = cur
            elif not torch.equal(ref, cur): stable = False
        outs.append((ref, stable))
    lib.cfp_debug_set(28, 0)
    v, sp = ctypes.c_int(0), ctypes.c_int(0)
    lib.cfp_conv2d_plan(M, Cout, KH * KW * Cin, KH, st, hip.F32X3, Ho * Wo if (a[26] & 1) else 0, B, ctypes.byref(v), ctypes.byref(sp))
    same = torch.equal(outs[0][0], outs[1][0])
    flag = "" if (outs[0][1] and outs[1][1] and same) else "   <<<<<<"
    print(f"M={M:7d} N={Cout:4d} K={KH*KW*Cin:5d} k={KH} ln={bool(a[23])} flags={a[26]} plan v{v.value}/s{sp.value}: pipelined stable {outs[0][1]}, plain stable {outs[1][1]}, equal {same}{flag}")
