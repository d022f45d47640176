"""Seeded synthetic inputs of the shapes the reference's loaders produce.

There is no dataset on either box, so tests and the benchmark feed the model the same kind of
batch `NYUV2.__getitem__` assembles (`/root/reference/src/dataloader/nyu.py:152-195`): an
ImageNet-normalised RGB image (nyu.py:269), a centred grid of ToF zones
(`src/utils/dataloader.py:93-103,121-123`), 16 depth samples per zone on mu +- 3 sigma
(dataloader.py:74-79), a zone validity mask and the integer `patch_info`.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .geometry import centered_zone_rects, collate_patch_info, patch_info_from_rect_data, sample_points_from_hist

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
SEED = 117010053  # train.py:218


def make_inputs(batch: int, height: int = 480, width: int = 640, zone_num: int = 8, zone_px: int = 56,
                seed: int = SEED, drop_hist: float = 0.0, zone_sample_num: int = 16,
                rect_shift: Tuple[int, int] = (0, 0), image_hw: Tuple[int, int] = (480, 640)) -> Dict:
    """A batch dict shaped like `train.py:104-115` builds it (all tensors on the CPU).

    rect_shift moves the whole zone grid (pixels); a shift that makes it overhang the image
    exercises the pad>0 path of `fusion.py:112-118`."""
    rng = np.random.default_rng(seed)
    rgb = rng.random((batch, 3, height, width), dtype=np.float32)
    rgb = (rgb - IMAGENET_MEAN[None, :, None, None]) / IMAGENET_STD[None, :, None, None]
    rects = centered_zone_rects(height, width, zone_num, zone_px)
    rects[:, [0, 2]] += rect_shift[0]
    rects[:, [1, 3]] += rect_shift[1]
    Z = zone_num * zone_num
    hist, masks, infos = [], [], []
    for _ in range(batch):
        mu = rng.uniform(0.5, 4.0, size=Z).astype(np.float32)
        sigma = rng.uniform(0.02, 0.2, size=Z).astype(np.float32)
        m = np.ones(Z, dtype=bool)
        if drop_hist > 0:
            idx = rng.choice(Z, int(Z * drop_hist), replace=False)
            m[idx] = False
        hist.append(sample_points_from_hist(np.stack([mu, sigma], 1), m, zone_sample_num))
        masks.append(m)
        infos.append(patch_info_from_rect_data(rects, image_hw))
    pi = collate_patch_info(infos)
    patch_info = {s: {k: torch.from_numpy(v) for k, v in pi[s].items()} for s in (4, 8, 16)}
    patch_info["zone_num"] = torch.from_numpy(pi["zone_num"])
    return {
        "rgb": torch.from_numpy(np.ascontiguousarray(rgb)),
        "additional": {
            "hist_data": torch.from_numpy(np.stack(hist)),
            "rect_data": torch.from_numpy(np.stack([rects] * batch)),
            "mask": torch.from_numpy(np.stack(masks)),
            "patch_info": patch_info,
        },
    }


def make_img_features(batch: int, height: int = 480, width: int = 640, seed: int = SEED + 1):
    """Stand-in encoder outputs (five maps, 16/40/56/136/232 channels at 1/2 .. 1/32): used to pin
    everything *after* the RGB encoder against the reference, whose encoder cannot run here."""
    rng = np.random.default_rng(seed)
    feats = []
    for i, c in enumerate((16, 40, 56, 136, 232)):
        s = 2 ** (i + 1)
        feats.append(torch.from_numpy(rng.standard_normal((batch, c, height // s, width // s), dtype=np.float32)))
    return feats


def to_device(input_data: Dict, device) -> Dict:
    """Move what `train.py:104-112` moves; patch_info stays on the host."""
    add = input_data["additional"]
    return {
        "rgb": input_data["rgb"].to(device),
        "additional": {
            "hist_data": add["hist_data"].to(device),
            "rect_data": add["rect_data"].to(device),
            "mask": add["mask"].to(device),
            "patch_info": add["patch_info"],
        },
    }


def make_depth(height: int, width: int, seed: int = SEED, holes: float = 0.0, quantise_mm: bool = False) -> np.ndarray:
    """Synthetic ground-truth depth [H, W] f32 for the ToF simulation: a tilted plane plus two boxes and mild noise
    (so zones see one or two depth clusters), optionally with a fraction of invalid (zero) pixels in blobs."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    d = 0.8 + 2.2 * x / width + 0.7 * y / height
    for _ in range(2):
        cy, cx = rng.integers(0, height), rng.integers(0, width)
        hh, ww = rng.integers(40, 160), rng.integers(40, 200)
        d[max(0, cy - hh):cy + hh, max(0, cx - ww):cx + ww] -= rng.uniform(0.3, 0.9)
    d += rng.normal(0.0, 0.01, d.shape)
    d = np.clip(d, 0.05, 9.5)
    if holes > 0:
        m = rng.random((height // 16 + 1, width // 16 + 1)) < holes
        m = np.kron(m, np.ones((16, 16), dtype=bool))[:height, :width]
        d[m] = 0.0
    if quantise_mm:          # NYU ground truth: uint16 millimetres / 1000 in float32
        return (np.round(d * 1000.0).astype(np.float32) / np.float32(1000.0)).astype(np.float32)
    return d.astype(np.float32)


def make_eval_pair(height: int, width: int, pred_h: int, pred_w: int, seed: int, holes: float = 0.0, noise: float = 0.1):
    """(gt [H,W], pred [pred_h,pred_w]) float32 for the metric tests: the prediction is the ground truth sampled at
    model resolution times log-normal noise, with a few out-of-range / non-finite pixels the protocols must clamp."""
    gt = make_depth(height, width, seed=seed, holes=holes)
    full = make_depth(height, width, seed=seed, holes=0.0)
    rng = np.random.default_rng(seed + 1000)
    ys = np.linspace(0, height - 1, pred_h).round().astype(int)
    xs = np.linspace(0, width - 1, pred_w).round().astype(int)
    pred = full[np.ix_(ys, xs)] * np.exp(rng.normal(0.0, noise, (pred_h, pred_w)))
    pred = pred.astype(np.float32)
    idx = rng.integers(0, pred.size, 24)
    pred.flat[idx[:8]] = 0.0
    pred.flat[idx[8:16]] = 37.5
    pred.flat[idx[16:20]] = np.inf
    pred.flat[idx[20:]] = -1.0
    return gt, pred
