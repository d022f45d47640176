"""NYU training augmentation on the device: host-side mirror of `DataLoadPreprocess.random_crop` / `train_preprocess` /
`augment_image` + `ToTensor` (`src/dataloader/nyu.py:128-136,204-245,266-285`).

The random numbers are drawn on the host with python `random` / `np.random` in the reference's order (so a seeded run picks
the same crops, flips and jitters); the pixels are touched once, by `cfp_nyu_augment` (`csrc/augment.hip`).  The random
rotation that precedes the crop (nyu.py:121-124: PIL `Image.rotate`, bilinear for the image, nearest for the depth) is
`draw_rotation` + `rotate` (`cfp_nyu_rotate`, byte-exact with Pillow).  There is no CPU implementation here.
"""
from __future__ import annotations

import ctypes as C
import random
from typing import Sequence, Tuple

import numpy as np
import torch

from . import hip

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def draw_params(H0: int, W0: int, H: int, W: int):
    """One sample's draws in the reference's order (nyu.py:209-210, 217, 223, 231, 235, 239)."""
    x0 = random.randint(0, W0 - W)
    y0 = random.randint(0, H0 - H)
    flip = random.random() > 0.5
    do_aug = random.random() > 0.5
    gamma, brightness, colors = 1.0, 1.0, np.ones(3)
    if do_aug:
        gamma = random.uniform(0.9, 1.1)
        brightness = random.uniform(0.75, 1.25)
        colors = np.random.uniform(0.9, 1.1, size=3)
    return x0, y0, flip, do_aug, gamma, brightness, colors


def draw_rotation(degree: float) -> float:
    """nyu.py:122 -- drawn BEFORE the crop / flip / jitter draws of `draw_params`."""
    return (random.random() - 0.5) * 2 * degree


def rotate_matrix(angle_deg: float, w: int, h: int):
    """Pillow's `Image.rotate`: the destination->source affine matrix of a rotation by `angle_deg` (counter-clockwise) about
    the image centre, in float64 with Pillow's own rounding (cos / sin rounded to 15 decimals)."""
    import math
    angle = angle_deg % 360.0
    if angle in (90.0, 180.0, 270.0):
        raise NotImplementedError("multiples of 90 degrees take Pillow's transpose path; nyu.py draws |angle| <= --degree (2.5)")
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = w / 2, h / 2
    m[2], m[5] = m[0] * -cx + m[1] * -cy + m[2], m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return m


def rotate(rgb_u8: torch.Tensor, depth_mm: torch.Tensor, angles: Sequence[float]) -> Tuple[torch.Tensor, torch.Tensor]:
    """rgb_u8 [B,H,W,3] uint8, depth_mm [B,H,W] uint16 (int16 storage) on the device, one angle (degrees) per sample ->
    rotated copies (same size, zero fill).  An angle of exactly 0 is a copy, like Pillow's."""
    assert rgb_u8.is_cuda and rgb_u8.dtype == torch.uint8 and rgb_u8.is_contiguous() and rgb_u8.shape[-1] == 3
    B, H, W, _ = rgb_u8.shape
    assert depth_mm.is_cuda and depth_mm.element_size() == 2 and tuple(depth_mm.shape) == (B, H, W) and depth_mm.is_contiguous()
    assert len(angles) == B
    mats = torch.tensor([rotate_matrix(float(a), W, H) for a in angles], dtype=torch.float64).to(rgb_u8.device)
    rgb_out, dep_out = torch.empty_like(rgb_u8), torch.empty_like(depth_mm)
    hip.call("cfp_nyu_rotate", rgb_u8.data_ptr(), depth_mm.data_ptr(), rgb_out.data_ptr(), dep_out.data_ptr(), B, H, W, mats.data_ptr(),
             hip.current_stream())
    return rgb_out, dep_out


def augment(rgb_u8: torch.Tensor, depth_mm: torch.Tensor, params: Sequence[tuple], H: int, W: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """rgb_u8 [B,H0,W0,3] uint8, depth_mm [B,H0,W0] uint16 (as int16 storage) on the device, params = one `draw_params`
    tuple per sample -> (image [B,3,H,W] f32 normalised, depth [B,1,H,W] f32 metres)."""
    assert rgb_u8.is_cuda and rgb_u8.dtype == torch.uint8 and rgb_u8.is_contiguous() and rgb_u8.shape[-1] == 3
    B, H0, W0, _ = rgb_u8.shape
    assert depth_mm.is_cuda and depth_mm.element_size() == 2 and tuple(depth_mm.shape) == (B, H0, W0) and depth_mm.is_contiguous()
    dev = rgb_u8.device
    pi = torch.tensor([[p[0], p[1], int(p[2]), int(p[3])] for p in params], dtype=torch.int32).to(dev)
    pf = torch.tensor([[p[4], p[5]] for p in params], dtype=torch.float32).to(dev)
    pc = torch.tensor(np.stack([np.asarray(p[6], dtype=np.float64) for p in params]), dtype=torch.float64).to(dev)
    img = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
    dep = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
    mean, std = (C.c_float * 3)(*MEAN), (C.c_float * 3)(*STD)
    hip.call("cfp_nyu_augment", rgb_u8.data_ptr(), depth_mm.data_ptr(), B, H0, W0, pi.data_ptr(), pf.data_ptr(), pc.data_ptr(), H, W, mean, std,
             img.data_ptr(), dep.data_ptr(), hip.current_stream())
    return img, dep
