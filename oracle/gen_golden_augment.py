#!/usr/bin/env python3
"""Golden vectors for the NYU augmentation: the reference's own `random_crop` / `train_preprocess` / `augment_image`
(`/root/reference/src/dataloader/nyu.py:204-245`) on seeded synthetic images with seeded draws.  Build container only.

`src.dataloader.nyu` imports torchvision, h5py and matplotlib, none of which is installed; they are only IMPORTED by the
methods used here, so import-only stubs stand in for them (no arithmetic comes from a stub).  The fixture stores the recorded
draws and the un-normalised outputs (the Normalize step is torchvision's and is not part of this pin).

    python oracle/gen_golden_augment.py        -> tests/golden/augment.npz
"""
import os
import random
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def source(seed, H0=456, W0=608):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (H0, W0, 3), dtype=np.uint8), rng.integers(0, 10000, (H0, W0), dtype=np.uint16)


def main():
    for name in ("torchvision", "h5py", "matplotlib", "matplotlib.pyplot", "matplotlib.patches", "timm"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = types.SimpleNamespace(Compose=lambda x: x, Normalize=lambda **k: None)
    sys.modules["matplotlib.patches"].Rectangle = object
    sys.modules["matplotlib"].patches = sys.modules["matplotlib.patches"]
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.path.insert(0, REF)
    sys.argv = ["gen_golden_augment"]
    from src.dataloader.nyu import DataLoadPreprocess
    obj = object.__new__(DataLoadPreprocess)
    out, H, W = {}, 416, 544
    for case, seed in enumerate((11, 12, 13, 14, 15, 16)):
        rgb, dmm = source(100 + seed)
        image = np.array(rgb, dtype=np.float32) / 255.
        depth = np.expand_dims(np.array(dmm, dtype=np.float32) / 1000.0, axis=2)
        random.seed(seed); np.random.seed(seed)
        img, dep = obj.random_crop(image, depth, H, W)
        img, dep = obj.train_preprocess(img, dep)
        # replay the same generators to record what was drawn (nyu.py:209-210,217,223,231,235,239)
        random.seed(seed); np.random.seed(seed)
        x0 = random.randint(0, image.shape[1] - W); y0 = random.randint(0, image.shape[0] - H)
        flip = random.random() > 0.5
        do_aug = random.random() > 0.5
        gamma = brightness = 1.0
        colors = np.ones(3)
        if do_aug:
            gamma = random.uniform(0.9, 1.1); brightness = random.uniform(0.75, 1.25); colors = np.random.uniform(0.9, 1.1, size=3)
        out[f"c{case}.params"] = np.array([x0, y0, int(flip), int(do_aug), gamma, brightness, *colors], dtype=np.float64)
        out[f"c{case}.img"] = np.asarray(img, dtype=np.float32)[::8, ::8].copy()        # [H/8, W/8, 3] un-normalised
        out[f"c{case}.dep"] = np.asarray(dep, dtype=np.float32)[::8, ::8, 0].copy()
        print(case, seed, x0, y0, flip, do_aug, round(gamma, 4), round(brightness, 4), colors.round(4), img.dtype)
    out["seeds"] = np.array([11, 12, 13, 14, 15, 16])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "augment.npz"), **out)


if __name__ == "__main__":
    main()
