#!/usr/bin/env python3
"""fp16 error of the engine for one WEIGHT FAMILY (cfpnet_amd.weights.FAMILIES; BatchNorm statistics calibrated for the kaiming ones,
tests/helpers.calibrate_bn), batch 8 at 480x640, against the float32 ENGINE on the same inputs (it equals the CPU oracle to 2e-6):

  * end to end per seed (rel-L1 of pred, worst image) for a list of engine switches (environment variables read at construction),
  * per-tap table (encoder taps, decoder stages, fusion blocks, unet, ram),
  * weights-only ablation by parameter group (float32 engine, one group's weights rounded to the 16-bit format).

    python tools/precision_family.py kaiming [--seeds 97 4242] [--env CFP_WEIGHTS2=1 CFP_HEAD_HILO=10] [--groups] [--taps]
"""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("family")
ap.add_argument("--seeds", type=int, nargs="*", default=[97, synthetic.SEED, 4242])
ap.add_argument("--env", nargs="*", default=[])
ap.add_argument("--groups", action="store_true")
ap.add_argument("--taps", action="store_true")
ap.add_argument("--acts", action="store_true", help="activation-rounding ablation: float32 engine with ONE group of encoder tensors rounded to the 16-bit format")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--dtype", default="f16")
a = ap.parse_args()
DT = {"f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers), family=a.family)
if a.family != "uniform":
    from helpers import calibrate_bn
    sd = calibrate_bn(sd, layers)


def rel(a_, b_):
    a_, b_ = a_.double().cpu().numpy(), b_.double().cpu().numpy()
    return float(np.abs(a_ - b_).sum() / max(np.abs(a_).sum(), 1e-30))


inps = {s: synthetic.to_device(synthetic.make_inputs(a.batch, 480, 640, 8, 56, seed=s, drop_hist=0.1 * (i % 3)), "cuda:0") for i, s in enumerate(a.seeds)}
e32 = Engine(sd, layer_names=layers, dtype=torch.float32)
ref, ref_taps = {}, {}
for s, inp in inps.items():
    t = {} if a.taps else None
    ref[s] = e32.forward(inp, taps=t)[1].clone()
    ref_taps[s] = t
print(f"family {a.family}: pred mean {float(ref[a.seeds[0]].mean()):.3f} std {float(ref[a.seeds[0]].std()):.3f}")
# conditioning of the network itself: the FLOAT32 engine on the same image rounded once to the 16-bit format (one perturbation of
# relative size 2^-12 / 2^-9 at the very first tensor, nothing else changed)
for s_, inp in inps.items():
    pert = {"rgb": inp["rgb"].to(DT).float(), "additional": inp["additional"]}
    pp = e32.forward(pert)[1]
    per = [rel(ref[s_][b], pp[b]) for b in range(pp.shape[0])]
    print(f"  input-rounding sensitivity ({a.dtype} image, float32 everything else) seed {s_}: rel-L1 {rel(ref[s_], pp):.3e} (worst image {max(per):.3e})")


def run(envs):
    old = {}
    for kv in envs:
        k, v = kv.split("=")
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        eng = Engine(sd, layer_names=layers, dtype=DT)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    out = []
    for s, inp in inps.items():
        t = {} if a.taps else None
        p = eng.forward(inp, taps=t)[1]
        torch.cuda.synchronize()
        per = [rel(ref[s][b], p[b]) for b in range(p.shape[0])]
        out.append((s, rel(ref[s], p), max(per)))
        if a.taps and not envs:
            for k in t:
                if torch.is_tensor(t[k]) and k in ref_taps[s] and t[k].shape == ref_taps[s][k].shape:
                    print(f"    tap {k:40s} rel-L1 {rel(ref_taps[s][k], t[k]):.3e}")
    del eng
    torch.cuda.empty_cache()
    return out


for envs in [[]] + [[e] for e in a.env] + ([a.env] if len(a.env) > 1 else []):
    r = run(envs)
    print(f"{a.dtype} {' '.join(envs) or 'default':40s}: " + "  ".join(f"seed {s}: {x:.3e} (worst {w:.3e})" for s, x, w in r), flush=True)

if a.groups:
    GROUPS = {
        "encoder.stem+stage0-2": lambda k: k.startswith(("img_encoder.conv0", "img_encoder.conv1", "img_encoder.conv2")),
        "encoder.stage3-5 (IR) pointwise": lambda k: k.startswith(("img_encoder.conv3", "img_encoder.conv4")) and "conv_dw" not in k,
        "encoder.stage3-5 (IR) depthwise": lambda k: k.startswith(("img_encoder.conv3", "img_encoder.conv4")) and "conv_dw" in k,
        "decoder.up1-4": lambda k: k.startswith("decoder.up"),
        "decoder.conv1-4": lambda k: k.startswith(("decoder.conv4", "decoder.conv3", "decoder.conv2", "decoder.conv1")),
        "fusion LoFTR linears": lambda k: k.startswith("decoder.cross_atten") and k.endswith(("q_proj.weight", "k_proj.weight", "v_proj.weight", "merge.weight", "mlp.0.weight", "mlp.2.weight")),
        "fusion DAPM 3x3": lambda k: k.startswith("decoder.cross_atten") and ".transformer_path.conv" in k,
        "fusion LKPM": lambda k: k.startswith("decoder.cross_atten") and ".large_kernel_path." in k,
        "fusion GSA sr": lambda k: k.startswith("decoder.cross_atten") and ".gsa.sr." in k,
        "decoder.conv0": lambda k: k.startswith("decoder.conv0"),
        "depth_head.conv3x3": lambda k: k.startswith("depth_head.conv3x3"),
        "conv_out": lambda k: k.startswith("conv_out"),
    }

    def roundable(k, v):
        return (torch.is_tensor(v) and v.is_floating_point() and v.dim() >= 2 and "positional" not in k and ".se." not in k
                and "regressor" not in k and "conv1x1" not in k and "hist_encoder" not in k)
    s0 = a.seeds[0]
    tot = 0.0
    for g, sel in GROUPS.items():
        e32.load_state_dict({k: (v.to(DT).float() if roundable(k, v) and sel(k) else v) for k, v in sd.items()})
        r = rel(ref[s0], e32.forward(inps[s0])[1])
        tot += r * r
        print(f"  weights rounded (nearest) in {g:36s}: {r:.3e}")
    print(f"  root-sum-square {tot ** 0.5:.3e}")
    e32.load_state_dict({k: (v.to(DT).float() if roundable(k, v) else v) for k, v in sd.items()})
    print(f"  ALL weights rounded (nearest), f32 activations: {rel(ref[s0], e32.forward(inps[s0])[1]):.3e}")

if a.acts:
    from cfpnet_amd import spec as _spec
    nb = len(_spec.ENC_BLOCKS)
    kinds = [b.kind for b in _spec.ENC_BLOCKS]
    groups = {"input image": None, "stem": "stem"}
    st = 0
    names = {}
    for bi, b in enumerate(_spec.ENC_BLOCKS):
        stage = b.prefix.rsplit(".", 1)[0]
        names.setdefault(stage, []).append(bi)
    for stage, bis in names.items():
        groups[f"{stage} block outputs"] = ",".join(f"enc{bi}:{a.dtype}" for bi in bis)
        groups[f"{stage} expanded (mid)"] = ",".join(f"enc{bi}.mid:{a.dtype}" for bi in bis)
        if kinds[bis[0]] == "ir":
            groups[f"{stage} depthwise out"] = ",".join(f"enc{bi}.dw:{a.dtype}" for bi in bis)
    groups["stem"] = f"stem:{a.dtype}"
    s0 = a.seeds[0]
    tot = 0.0
    for gname, spec_ in groups.items():
        if spec_ is None:
            continue
        os.environ["CFP_DEBUG_ROUND"] = spec_
        e = Engine(sd, layer_names=layers, dtype=torch.float32)
        r = rel(ref[s0], e.forward(inps[s0])[1])
        tot += r * r
        print(f"  activations rounded to {a.dtype} in {gname:36s}: {r:.3e}", flush=True)
        del e
    os.environ.pop("CFP_DEBUG_ROUND", None)
    print(f"  root-sum-square over the encoder groups {tot ** 0.5:.3e}")
