#!/usr/bin/env python3
"""Per-tensor fidelity of the 16-bit training gradients against the float32 ones at the benched shard (16 crops of 416x544, 6x6 zones,
34 % dropped): cosine and rms ratio of every parameter-gradient tensor, each tensor's share of the squared error of the WHOLE
gradient (what the full-gradient cosine is made of), grouped by module.  Writes a markdown table.

    python tools/train_fidelity_table.py [--out profiles/r3_train_fidelity.md] [--batch 16]
"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import spec, synthetic, weights
from cfpnet_amd.train_model import TrainNet

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/train_fidelity.md")
ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
B, H, W = a.batch, 416, 544
layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
inp = synthetic.to_device(synthetic.make_inputs(B, H, W, 6, 64, seed=5, drop_hist=0.34), "cuda:0")
target = torch.from_numpy(np.stack([synthetic.make_depth(H, W, seed=50 + i, holes=0.1) for i in range(B)]))[:, None].cuda()
offs = {"cross_atten3": (3, 5), "cross_atten2": (7, 2), "cross_atten1": (11, 20)}
grads, losses = {}, {}
for name, dt in (("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
    net = TrainNet(sd, layers, "cuda:0", dtype=dt)
    loss, _, _ = net.forward_backward(inp, target, target > 1e-3, pos_offsets=offs)
    torch.cuda.synchronize()
    grads[name] = {k: v.double().cpu() for k, v in net.grads().items()}
    losses[name] = float(loss)
    del net
    torch.cuda.empty_cache()


def group_of(k):
    p = k.split(".")
    if p[0] == "img_encoder":
        return "img_encoder." + (p[1] if p[1] not in ("conv3",) else p[1] + "." + p[2])
    if p[0] == "decoder" and p[1].startswith("cross_atten") and p[2] != "layers":
        return f"decoder.{p[1]}.positional tables"
    if p[0] == "decoder" and p[1].startswith("cross_atten"):
        kind = "hist2image" if int(p[3]) in (0, 3) else ("combine1" if int(p[3]) in (1, 4) else "image")
        return f"decoder.{p[1]}.{kind}"
    return ".".join(p[:2])


lines = [f"# Gradient fidelity of mixed-precision training at the benched shard ({B} crops of {H}x{W}, float32 = reference)", "",
         f"loss: f32 {losses['f32']:.6f}, fp16 {losses['f16']:.6f} ({abs(losses['f16'] - losses['f32']) / losses['f32']:.2e} rel), "
         f"bf16 {losses['bf16']:.6f} ({abs(losses['bf16'] - losses['f32']) / losses['f32']:.2e} rel)", ""]
ref = grads["f32"]
keys = sorted(ref)
for name in ("f16", "bf16"):
    g = grads[name]
    tot_err = sum(float(((g[k] - ref[k]) ** 2).sum()) for k in keys)
    tot_ref = sum(float((ref[k] ** 2).sum()) for k in keys)
    dot = sum(float((g[k] * ref[k]).sum()) for k in keys)
    gn = sum(float((g[k] ** 2).sum()) for k in keys)
    lines += [f"## {name}: full-gradient cosine {dot / (gn * tot_ref) ** 0.5:.4f}, |g - g32| / |g32| = {(tot_err / tot_ref) ** 0.5:.4f}, {len(keys)} tensors", ""]
    rows = []
    for k in keys:
        e = float(((g[k] - ref[k]) ** 2).sum()); r = float((ref[k] ** 2).sum()); gg = float((g[k] ** 2).sum())
        cos = float((g[k] * ref[k]).sum()) / max((gg * r) ** 0.5, 1e-300)
        rows.append((e / tot_err, k, cos, (gg / max(r, 1e-300)) ** 0.5, (r / tot_ref), ref[k].numel()))
    by_group = {}
    for share, k, cos, ratio, norm_share, n in rows:
        d = by_group.setdefault(group_of(k), [0.0, 0.0, 0, []])
        d[0] += share; d[1] += norm_share; d[2] += 1; d[3].append(cos)
    lines += ["| module group | tensors | share of the squared error | share of the squared norm | median cosine | min cosine |", "|---|---|---|---|---|---|"]
    for gname, (share, ns, cnt, coss) in sorted(by_group.items(), key=lambda kv: -kv[1][0]):
        lines.append(f"| {gname} | {cnt} | {100 * share:.1f} % | {100 * ns:.1f} % | {float(np.median(coss)):.4f} | {min(coss):.4f} |")
    lines += ["", "| tensor (30 largest error shares) | elements | error share | norm share | cosine | rms ratio |", "|---|---|---|---|---|---|"]
    for share, k, cos, ratio, norm_share, n in sorted(rows, reverse=True)[:30]:
        lines.append(f"| {k} | {n} | {100 * share:.2f} % | {100 * norm_share:.2f} % | {cos:.4f} | {ratio:.3f} |")
    coss = np.array([r[2] for r in rows])
    lines += ["", f"cosine per tensor: median {np.median(coss):.4f}, 10th percentile {np.percentile(coss, 10):.4f}, min {coss.min():.4f}; "
              f"tensors below 0.9: {(coss < 0.9).sum()}, below 0.5: {(coss < 0.5).sum()}", ""]
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
open(a.out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
