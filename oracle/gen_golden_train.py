#!/usr/bin/env python3
"""Golden vectors for the TRAINING step: the reference's own modules in `.train()` (batch-statistics BatchNorm), its
SILogLoss and `loss.backward()` (train.py:119-125), on stand-in encoder features (timm is not installed, so the RGB encoder
cannot run; everything after it can -- same arrangement as oracle/gen_golden.py, whose import shim this script reuses).

Build container only.  Writes tests/golden/train_step.npz: the loss, a slice of the prediction, the recorded positional
draws, and for EVERY decoder / histogram-encoder / head parameter the gradient's (mean, |mean|, rms) plus the full gradient
of a dozen small tensors.  `tests/test_oracle_golden.py` pins the oracle's BN_TRAIN + autograd path against it.

    python oracle/gen_golden_train.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden as GG  # noqa: E402
from cfpnet_amd import spec, synthetic, weights  # noqa: E402

CASE = dict(B=2, H=256, W=320, zn=3, zpx=64, seed=303, drop=0.25)
FULL = ["decoder.conv4.bias", "decoder.up1._net.1.weight", "decoder.cross_atten3.layers.0.norm1.weight", "decoder.cross_atten3.layers.1.large_kernel_path.bn1.weight",
        "decoder.cross_atten2.layers.2.gsa.norm.bias", "decoder.cross_atten1.layers.1.transformer_path.bn2.weight", "decoder.cross_atten1.layers.0.norm2.bias",
        "hist_encoder.hist_extractor3.pointnet_encoder.bn3.weight", "hist_encoder.hist_extractor1.pointnet_encoder.conv1.weight",
        "depth_head.regressor.4.bias", "conv_out.0.bias", "decoder.conv0.bias"]


def main():
    ref = GG.import_reference()
    rcfg, Decoder, DepthRegression, HistogramEncoder, SILogLoss = ref[0], ref[1], ref[2], ref[3], ref[4]
    layers = spec.COMBINE1_LAYERS
    args = rcfg.args
    args.attention_layer, args.change_embedding, args.no_skip_inside = list(layers), True, False
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    dec = GG.load_into(Decoder(num_classes=128), "decoder", sd).train()
    he = GG.load_into(HistogramEncoder(), "hist_encoder", sd).train()
    dh = GG.load_into(DepthRegression(128, dim_out=256, norm="linear"), "depth_head", sd).train()
    conv_out = torch.nn.Sequential(torch.nn.Conv2d(128, 256, 1), torch.nn.Softmax(dim=1))
    GG.load_into(conv_out, "conv_out", sd).train()
    c = CASE
    inputs = synthetic.make_inputs(c["B"], c["H"], c["W"], c["zn"], c["zpx"], seed=c["seed"], drop_hist=c["drop"])
    feats = synthetic.make_img_features(c["B"], c["H"], c["W"], seed=c["seed"] + 1)
    target = torch.from_numpy(np.stack([synthetic.make_depth(c["H"], c["W"], seed=c["seed"] + 5 + i, holes=0.1) for i in range(c["B"])]))[:, None]
    draws, real = [], torch.randint

    def rec(*a, **k):
        v = real(*a, **k)
        draws.append(int(v))
        return v
    torch.randint = rec
    try:
        add = inputs["additional"]
        hist_features = he(add["hist_data"].unsqueeze(-1))
        unet = dec(feats, hist_features, rect_data=add["rect_data"], mask=add["mask"], patch_info=add["patch_info"], rgb=inputs["rgb"])
        widths, ram = dh(unet)
        prob = conv_out(ram)
        bw = torch.nn.functional.pad((10.0 - 1e-3) * widths, (1, 0), mode="constant", value=1e-3)
        edges = torch.cumsum(bw, dim=1)
        centers = 0.5 * (edges[:, :-1] + edges[:, 1:])
        pred = torch.sum(prob * centers.view(*centers.shape, 1, 1), dim=1, keepdim=True)
    finally:
        torch.randint = real
    mask = target > 1e-3
    loss = SILogLoss()(torch.clip(pred, 1e-3), target, mask=mask.to(torch.bool), interpolate=True)        # train.py:121-123
    loss.backward()
    out = {"loss": np.float64(loss.item()), "pred_slice": pred.detach()[:, :, ::4, ::4].numpy(), "draws": np.array(draws, dtype=np.int64)}
    stats, names = [], []
    for prefix, mod in (("decoder", dec), ("hist_encoder", he), ("depth_head", dh), ("conv_out", conv_out)):
        for n, p in mod.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.double()
            names.append(f"{prefix}.{n}")
            stats.append([float(g.mean()), float(g.abs().mean()), float((g * g).mean().sqrt()), float(g.abs().max())])
            if f"{prefix}.{n}" in FULL:
                out["grad." + f"{prefix}.{n}"] = p.grad.numpy().astype(np.float32)
    out["grad_stats"] = np.array(stats)
    meta = dict(CASE, names=names, full=FULL, running=dict(
        up1=dec.up1._net[1].running_mean[:8].tolist(), hist=he.hist_extractor1.pointnet_encoder.bn1.running_var[:8].tolist()))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_step.npz"), **out)
    print("loss", loss.item(), "params with grad", len(names), "draws", draws)


if __name__ == "__main__":
    main()
