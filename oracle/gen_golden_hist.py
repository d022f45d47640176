#!/usr/bin/env python3
"""Golden vectors for the ToF simulation `get_hist_parallel` + `sample_point_from_hist_parallel`
(`/root/reference/src/utils/dataloader.py:66-134`) by running THE REFERENCE ITSELF on seeded synthetic depth maps.

Build container only (needs /root/reference).  Writes tests/golden/hist_sim.npz: per case the outputs (mu/sigma in
float64 as the reference produces them, validity mask, zone rectangles, the 16 sample depths); the inputs are
regenerated from the seed by `cfpnet_amd.synthetic.make_depth`.

    python oracle/gen_golden_hist.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cfpnet_amd import synthetic  # noqa: E402

# (name, mode, H, W, train_zone_num, seed, holes)
CASES = [("eval480", "online_eval", 480, 640, 8, 31, 0.0), ("eval480_holes", "online_eval", 480, 640, 8, 32, 0.35),
         ("train416", "train", 416, 544, 6, 33, 0.2), ("train416_z4", "train", 416, 544, 4, 34, 0.0),
         ("eval480_sparse", "online_eval", 480, 640, 8, 35, 0.97), ("eval480_mm", "online_eval", 480, 640, 8, 36, -1.0)]


def main():
    timm = types.ModuleType("timm")
    sys.modules.setdefault("timm", timm)
    sys.path.insert(0, REF)
    sys.argv = ["gen_golden_hist"]
    import src.utils.dataloader as rdl
    out = {}
    meta = []
    for name, mode, H, W, tzn, seed, holes in CASES:
        cfg = types.SimpleNamespace(mode=mode, train_zone_num=tzn, train_zone_random_offset=0, simu_max_distance=4.0,
                                    zone_sample_num=16, sample_uniform=True)
        # holes < 0: depth quantised to millimetres like the NYU PNGs (values sit exactly on bin edges)
        dep = torch.from_numpy(synthetic.make_depth(H, W, seed=seed, holes=max(holes, 0.0), quantise_mm=holes < 0))[None]
        rgb = torch.zeros(3, H, W)
        fh, fr, mask = rdl.get_hist_parallel(rgb, dep, cfg)
        pts = rdl.sample_point_from_hist_parallel(fh, mask, cfg)
        out[name + ".fh"] = fh.numpy().astype(np.float64)
        out[name + ".fr"] = fr.numpy().astype(np.float32)
        out[name + ".mask"] = mask.numpy().astype(np.uint8)
        out[name + ".pts"] = pts.numpy().astype(np.float32)
        out[name + ".w0"] = torch.linspace(1, 0, 16).numpy()                # tensor_linspace's tables on THIS host
        out[name + ".w1"] = torch.linspace(0, 1, 16).numpy()
        meta.append(dict(name=name, mode=mode, H=H, W=W, train_zone_num=tzn, seed=seed, holes=holes, fh_dtype=str(fh.dtype)))
        print(name, "valid zones", int(mask.sum()), "/", mask.numel(), "mu[:3]", fh[:3, 0].tolist())
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hist_sim.npz"), **out)
    # the non-uniform branch (argparse default: no --sample_uniform), dataloader.py:69-73: the reference's own samples for
    # the (mu, sigma, mask) of every case above, with some valid zones masked out like drop_hist does (nyu.py:155-158)
    icdf = {}
    import math
    for m in meta:
        n = m["name"]
        cfg = types.SimpleNamespace(zone_sample_num=16, sample_uniform=False)
        fh, mask = torch.from_numpy(out[n + ".fh"]), torch.from_numpy(out[n + ".mask"].astype(bool))
        mask[1::5] = False
        pts = rdl.sample_point_from_hist_parallel(fh, mask, cfg)
        assert pts.dtype == torch.float32 and pts.shape == (mask.numel(), 16)
        icdf[n + ".mask"] = mask.numpy().astype(np.uint8)
        icdf[n + ".pts"] = pts.numpy()
    delta = 1e-3
    ppf = torch.Tensor(np.arange(delta, 1, (1 - 2 * delta) / 15).tolist())
    icdf["table"] = torch.erfinv(2 * ppf - 1).numpy()                         # float32 erfinv on THIS host
    icdf["ppf"] = ppf.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hist_sim_icdf.npz"), **icdf)
    print("icdf: table", icdf["table"][:3], "pts[0]", icdf[meta[0]["name"] + ".pts"][0, :4])


if __name__ == "__main__":
    main()
