#!/usr/bin/env python3
"""Golden vectors for the random rotation of the NYU loader (`/root/reference/src/dataloader/nyu.py:121-124,200-202`): the
reference's own call sequence -- `image.rotate(angle, resample=Image.BILINEAR)` on an RGB image and
`depth.rotate(angle, resample=Image.NEAREST)` on a 16-bit ("I;16") depth image -- executed by Pillow (the reference's pinned
third-party dependency, installed in this image) on seeded synthetic images.  Build container only.

    python oracle/gen_golden_augment_rotate.py        -> tests/golden/augment_rotate.npz  (inputs, angles, Pillow's outputs)
"""
import os
import random

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = {}
    rng = np.random.default_rng(2024)
    cases = [(57, 76, None), (57, 76, None), (40, 40, None), (64, 48, 33.3), (57, 76, 0.0)]
    for i, (h, w, fixed) in enumerate(cases):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        dep = rng.integers(0, 10000, (h, w), dtype=np.uint16)
        random.seed(40 + i)
        angle = (random.random() - 0.5) * 2 * 2.5 if fixed is None else fixed          # nyu.py:122 with --degree 2.5
        im, dm = Image.fromarray(rgb, "RGB"), Image.fromarray(dep)
        assert dm.mode == "I;16"
        out[f"c{i}.rgb"], out[f"c{i}.dep"], out[f"c{i}.angle"] = rgb, dep, np.float64(angle)
        out[f"c{i}.rgb_rot"] = np.array(im.rotate(angle, resample=Image.BILINEAR))
        out[f"c{i}.dep_rot"] = np.array(dm.rotate(angle, resample=Image.NEAREST))
        print(i, (h, w), angle, int(out[f"c{i}.rgb_rot"].sum()), int(out[f"c{i}.dep_rot"].sum()))
    out["n"] = np.int64(len(cases))
    import PIL
    out["pillow_version"] = np.array(PIL.__version__)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "augment_rotate.npz"), **out)


if __name__ == "__main__":
    main()
