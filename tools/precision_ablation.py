#!/usr/bin/env python3
"""Which part of the 16-bit error is whose?  Exact ablations with the real engines (no emulation):

  full        16-bit engine end to end                                   (what bench.py reports)
  weights     f32 engine whose conv / linear weights were rounded to the 16-bit format first   -> weight rounding only
  encoder     16-bit RGB encoder, its five taps fed to the f32 decoder    -> everything the encoder contributes
  decoder     f32 encoder taps fed to the 16-bit decoder + head           -> everything after the encoder
  hist_f32    16-bit engine with the ToF histogram encoder outputs replaced by f32-engine values (via state: not available) -- skipped

All numbers are relative L1 of `pred` against the f32 engine on the same input (B=1, 480x640, bench seed).
    python tools/precision_ablation.py [--batch 1]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import spec, synthetic, weights  # noqa: E402
from cfpnet_amd.engine import Engine  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    return float(np.abs(a - b).sum() / np.abs(a).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    inp = synthetic.to_device(synthetic.make_inputs(a.batch), "cuda:0")
    e32 = Engine(sd, layer_names=layers, dtype=torch.float32)
    t32 = {}
    _, p32, _ = e32.forward(inp, taps=t32)
    feats32 = [t32[f"enc{i}"] for i in range(5)]
    for dt, name in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        e16 = Engine(sd, layer_names=layers, dtype=dt)
        t16 = {}
        _, p16, _ = e16.forward(inp, taps=t16)
        feats16 = [t16[f"enc{i}"] for i in range(5)]
        # weight rounding only: every >= 2-D floating tensor that the engine packs as a GEMM / conv operand
        sdw = {k: (v.to(dt).float() if (torch.is_tensor(v) and v.is_floating_point() and v.dim() >= 2 and "positional" not in k
                                         and ".se." not in k and "regressor" not in k and "conv1x1" not in k) else v)
               for k, v in sd.items()}
        ew = Engine(sdw, layer_names=layers, dtype=torch.float32)
        _, pw, _ = ew.forward(inp)
        _, pe, _ = e32.forward(inp, img_features=feats16)
        _, pd, _ = e16.forward(inp, img_features=feats32)
        torch.cuda.synchronize()
        print(f"{name}: full {rel(p32, p16):.3e} | weights-only {rel(p32, pw):.3e} | encoder-only {rel(p32, pe):.3e} | "
              f"decoder+head-only {rel(p32, pd):.3e}")
        del e16, ew


if __name__ == "__main__":
    main()
