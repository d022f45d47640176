#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; nothing here is imported on the GPU
box).  The reference is imported in place -- no reference source is copied -- behind a ~10-line
`timm` shim, because timm is not installed (the RGB encoder therefore cannot run; everything
after it can).  Parameters are the key-addressed deterministic tensors of
`cfpnet_amd/weights.py` and inputs come from `cfpnet_amd/synthetic.py`, so fixtures only hold
*outputs* (plus tiny input tables for the integer known-answer tests).

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz, *.json

What is captured:
  decoder_*.npz   reference `Decoder + HistogramEncoder + DepthRegression + conv_out + bin maths`
                  composed exactly as deltar.py:39-61 does, on stand-in encoder features
  geometry.json   patch_info_from_rect_data / sample_point_from_hist_parallel known answers
  misc.json       SILogLoss, compute_errors, config parse results
  manifest.json   key -> shape of the reference modules' state_dict
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
CFG = "configs/train_deltar_change_embedding_no_clip_grad_hist_encoder_optimized_10x_combine1.txt"


def import_reference():
    timm = types.ModuleType("timm")
    timm.create_model = lambda *a, **k: None
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    registry = types.ModuleType("timm.models.registry")
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    layers.DropPath = torch.nn.Identity
    registry.register_model = lambda f: f
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers,
                        "timm.models.registry": registry})
    sys.path.insert(0, REF)
    sys.argv = ["gen_golden", "@" + os.path.join(REF, CFG)]   # config.py parses argv at import
    import src.config as rcfg
    from src.models.decoder import Decoder, DepthRegression
    from src.models.encoder import HistogramEncoder
    from src.loss import SILogLoss
    from src.utils.metrics import compute_errors
    import src.utils.dataloader as rdl
    return rcfg, Decoder, DepthRegression, HistogramEncoder, SILogLoss, compute_errors, rdl


sys.path.insert(0, ROOT)
from cfpnet_amd import spec, synthetic, weights  # noqa: E402


def load_into(module: torch.nn.Module, prefix: str, sd):
    sub = {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}
    module.load_state_dict(sub, strict=True)
    return module.eval()


def summarize(t: torch.Tensor):
    t = t.detach().double()
    return [float(t.mean()), float(t.abs().mean()), float((t * t).mean().sqrt())]


def run_reference(ref, layer_names, inputs, feats, change_embedding=True, no_skip_inside=False,
                  pos_draws=None):
    rcfg, Decoder, DepthRegression, HistogramEncoder = ref[0], ref[1], ref[2], ref[3]
    args = rcfg.args
    args.attention_layer = list(layer_names)
    args.change_embedding = change_embedding
    args.no_skip_inside = no_skip_inside
    sd = weights.make_torch_state_dict(spec.model_manifest(layer_names))
    dec = load_into(Decoder(num_classes=128), "decoder", sd)
    he = load_into(HistogramEncoder(), "hist_encoder", sd)
    dh = load_into(DepthRegression(128, dim_out=256, norm="linear"), "depth_head", sd)
    conv_out = torch.nn.Sequential(torch.nn.Conv2d(128, 256, 1), torch.nn.Softmax(dim=1))
    load_into(conv_out, "conv_out", sd)

    taps = {}
    hooks = []
    for name in ("up1", "conv3", "cross_atten3", "up2", "conv2", "cross_atten2", "up3", "conv1",
                 "cross_atten1", "up4", "conv0"):
        hooks.append(getattr(dec, name).register_forward_hook(
            lambda m, i, o, n=name: taps.__setitem__("unet_out" if n == "conv0" else n, o.clone())))
    for fname in ("cross_atten1", "cross_atten2", "cross_atten3"):
        for li, layer in enumerate(getattr(dec, fname).layers):
            if layer_names[li] == "hist2image":
                continue   # the module output is the per-zone token block, not the token map
            def hk(m, i, o, n=f"decoder.{fname}.layers.{li}"):
                taps[n] = o.clone()   # later hist2image layers update the map in place
            hooks.append(layer.register_forward_hook(hk))

    # record torch.randint draws of the positional-encoding window (fusion.py:88-91)
    draws = []
    real_randint = torch.randint

    def rec_randint(*a, **k):
        if pos_draws is not None:
            v = torch.tensor([pos_draws[len(draws)]])
        else:
            v = real_randint(*a, **k)
        draws.append(int(v))
        return v
    torch.randint = rec_randint
    try:
        add = inputs["additional"]
        with torch.no_grad():
            hist_features = he(add["hist_data"].unsqueeze(-1))            # deltar.py:40
            unet = dec(feats, hist_features, rect_data=add["rect_data"], mask=add["mask"],
                       patch_info=add["patch_info"], rgb=inputs["rgb"])   # deltar.py:41-48
            widths, ram = dh(unet)                                        # deltar.py:50
            prob = conv_out(ram)                                          # deltar.py:51
            bw = (10.0 - 1e-3) * widths                                   # deltar.py:53-61
            bw = torch.nn.functional.pad(bw, (1, 0), mode="constant", value=1e-3)
            edges = torch.cumsum(bw, dim=1)
            centers = 0.5 * (edges[:, :-1] + edges[:, 1:])
            pred = torch.sum(prob * centers.view(*centers.shape, 1, 1), dim=1, keepdim=True)
    finally:
        torch.randint = real_randint
        for h in hooks:
            h.remove()
    for i, f in enumerate(hist_features):
        taps[f"hist{i}"] = f
    taps["ram"] = ram
    taps["widths"] = widths
    return edges, pred, prob, taps, draws


def decoder_case(ref, name, layer_names, B, H, W, zn, zpx, seed, drop=0.0, shift=(0, 0), full_pred=True,
                 change_embedding=True, no_skip_inside=False, pos_draws=None):
    inputs = synthetic.make_inputs(B, H, W, zn, zpx, seed=seed, drop_hist=drop, rect_shift=shift)
    feats = synthetic.make_img_features(B, H, W, seed=seed + 1)
    edges, pred, prob, taps, draws = run_reference(ref, layer_names, inputs, feats, change_embedding,
                                                   no_skip_inside, pos_draws)
    meta = dict(name=name, layer_names=list(layer_names), B=B, H=H, W=W, zone_num=zn, zone_px=zpx, seed=seed,
                drop_hist=drop, rect_shift=list(shift), change_embedding=change_embedding,
                no_skip_inside=no_skip_inside, pos_draws=draws, full_pred=full_pred,
                tap_stats={k: summarize(v) for k, v in taps.items()})
    arrays = dict(
        bin_edges=edges.numpy().astype(np.float32),
        pred=(pred if full_pred else pred[:, :, ::4, ::4]).numpy().astype(np.float32),
        prob_slice=prob[:, :, ::16, ::16].numpy().astype(np.float32),   # [B,256,H/32,W/32]
        unet_slice=taps["unet_out"][:, ::8, ::8, ::8].numpy().astype(np.float32),
        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
    )
    for fname in ("cross_atten1", "cross_atten2", "cross_atten3"):
        arrays[fname + "_slice"] = taps[fname][:, ::4, ::4, ::4].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f"decoder_{name}.npz"), **arrays)
    print(f"  {name}: pred mean {float(pred.mean()):.4f} std {float(pred.std()):.4f} draws {draws}")


def geometry_kats(rdl, rcfg):
    cases = {}
    from cfpnet_amd.geometry import centered_zone_rects
    rect_sets = {
        "eval_8x8_56": centered_zone_rects(480, 640, 8, 56),
        "train_6x6_64": centered_zone_rects(416, 544, 6, 64),
        "overhang_top_left": centered_zone_rects(480, 640, 8, 56) + np.array([-40, -110, -40, -110], np.float32),
        "overhang_bottom_right": centered_zone_rects(480, 640, 8, 56) + np.array([30, 100, 30, 100], np.float32),
        "zju_like_ragged": (centered_zone_rects(480, 640, 8, 56) * np.float32(1.03)).astype(np.float32),
        "offset_odd": centered_zone_rects(480, 640, 8, 56, offset=7),
        "zone_2x2": centered_zone_rects(480, 640, 8, 56).reshape(8, 8, 4)[3:5, 3:5].reshape(4, 4),
        "zone_4x4": centered_zone_rects(480, 640, 8, 56).reshape(8, 8, 4)[2:6, 2:6].reshape(16, 4),
    }
    for name, rects in rect_sets.items():
        pi = rdl.patch_info_from_rect_data(torch.from_numpy(rects))
        out = {"rects": rects.tolist(), "zone_num": pi["zone_num"]}
        for s in (4, 8, 16):
            out[str(s)] = {k: [int(v) for v in pi[s][k]] for k in ("pad_size", "patch_size", "index_wo_pad")}
        cases[name] = out
    # sample_point_from_hist_parallel, uniform branch
    rng = np.random.default_rng(5)
    ms = np.stack([rng.uniform(0.5, 4, 64), rng.uniform(0.02, 0.2, 64)], 1).astype(np.float32)
    mask = rng.random(64) > 0.3
    cfgns = types.SimpleNamespace(zone_sample_num=16, sample_uniform=True)
    fh = rdl.sample_point_from_hist_parallel(torch.from_numpy(ms), torch.from_numpy(mask), cfgns)
    cases["sample_points"] = {"mu_sigma": ms.tolist(), "mask": mask.tolist(),
                              "out_hex": [np.float32(v).tobytes().hex() for v in fh.numpy().reshape(-1)]}
    return cases


def misc_kats(ref):
    rcfg, SILogLoss, compute_errors = ref[0], ref[4], ref[5]
    rng = np.random.default_rng(11)
    pred = torch.from_numpy(rng.uniform(0.3, 9.0, (2, 1, 26, 34)).astype(np.float32))
    gt = torch.from_numpy(rng.uniform(0.0, 9.0, (2, 1, 52, 68)).astype(np.float32))
    mask = gt > 1.0
    loss = SILogLoss()(pred, gt, mask=mask, interpolate=True)
    loss_nomask = SILogLoss()(pred, gt[:, :, ::2, ::2].clamp(min=0.1), mask=None, interpolate=False)
    g = rng.uniform(0.5, 9.0, 5000).astype(np.float32)
    p = (g * rng.uniform(0.7, 1.4, 5000)).astype(np.float32)
    errs = {k: float(v) for k, v in compute_errors(g, p).items()}
    out = {"silog": {"seed": 11, "loss": float(loss), "loss_nomask": float(loss_nomask)},
           "compute_errors": {"seed": 11, "values": errs}}
    # config parse of the two shipped .txt files
    cfgs = {}
    for f in sorted(os.listdir(os.path.join(REF, "configs"))):
        if f.endswith(".txt"):
            ns = rcfg.parser.parse_args(["@" + os.path.join(REF, "configs", f)])
            cfgs[f] = {k: v for k, v in vars(ns).items()}
    out["configs"] = cfgs
    out["config_defaults"] = {k: v for k, v in vars(rcfg.parser.parse_args([])).items()}
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    rcfg, Decoder, DepthRegression, HistogramEncoder = ref[:4]

    # state-dict manifest of the reference modules
    args = rcfg.args
    man = {}
    for layers, tag in ((spec.COMBINE1_LAYERS, "combine1"), (spec.BASELINE_LAYERS, "baseline")):
        args.attention_layer = list(layers)
        mods = {"decoder": Decoder(num_classes=128), "hist_encoder": HistogramEncoder(),
                "depth_head": DepthRegression(128, dim_out=256),
                "conv_out": torch.nn.Sequential(torch.nn.Conv2d(128, 256, 1), torch.nn.Softmax(dim=1))}
        man[tag] = {f"{p}.{k}": list(v.shape) for p, m in mods.items() for k, v in m.state_dict().items()}
    json.dump(man, open(os.path.join(OUT, "manifest.json"), "w"), indent=0, sort_keys=True)

    json.dump(geometry_kats(ref[6], rcfg), open(os.path.join(OUT, "geometry.json"), "w"))
    json.dump(misc_kats(ref), open(os.path.join(OUT, "misc.json"), "w"), indent=1, sort_keys=True, default=str)

    S = synthetic.SEED
    print("decoder goldens:")
    decoder_case(ref, "eval480_b1", spec.COMBINE1_LAYERS, 1, 480, 640, 8, 56, S)
    decoder_case(ref, "eval480_b2_drop", spec.COMBINE1_LAYERS, 2, 480, 640, 8, 56, S + 10, drop=0.34, full_pred=False)
    decoder_case(ref, "train416_b1", spec.COMBINE1_LAYERS, 1, 416, 544, 6, 64, S + 20, drop=0.34)
    decoder_case(ref, "overhang_b1", spec.COMBINE1_LAYERS, 1, 480, 640, 8, 56, S + 30, shift=(-40, -110), full_pred=False)
    decoder_case(ref, "baseline_b1", spec.BASELINE_LAYERS, 1, 480, 640, 8, 56, S + 40, full_pred=False)
    decoder_case(ref, "noskip_stale_b1", spec.COMBINE1_LAYERS, 1, 480, 640, 8, 56, S + 50, drop=0.2, full_pred=False,
                 change_embedding=False, no_skip_inside=True)


if __name__ == "__main__":
    main()
