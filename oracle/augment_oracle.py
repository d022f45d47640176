"""CPU oracle for the NYU training augmentation -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of `DataLoadPreprocess.random_crop` / `train_preprocess` / `augment_image` and `ToTensor` + `Normalize`
(`/root/reference/src/dataloader/nyu.py:128-136,204-245,266-285`) for GIVEN random draws.  Only `tests/` may import it.

Parity status: crop / flip / gamma / brightness / colour / clip PINNED against the reference's own methods
(`oracle/gen_golden_augment.py` imports `src.dataloader.nyu` behind import-only stubs for torchvision / h5py / matplotlib
and replays seeded draws; fixture `tests/golden/augment.npz`).  The final `transforms.Normalize` is torchvision (absent
here): restated as its documented arithmetic `(x - mean) / std` in float32 -- that one line is UNPINNED.
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def augment(rgb_u8, depth_mm, x0, y0, flip, do_aug, gamma, brightness, colors, H, W, normalize=True):
    """rgb_u8 [H0,W0,3] uint8, depth_mm [H0,W0] uint16 -> (image [3,H,W] f32, depth [1,H,W] f32)."""
    image = np.array(rgb_u8, dtype=np.float32) / 255.                       # nyu.py:128
    depth = (np.array(depth_mm, dtype=np.float32) / 1000.0)[:, :, None]     # nyu.py:129-130
    image = image[y0:y0 + H, x0:x0 + W, :]                                  # random_crop, nyu.py:204-213
    depth = depth[y0:y0 + H, x0:x0 + W, :]
    if flip:                                                                # nyu.py:217-220
        image = image[:, ::-1, :].copy()
        depth = depth[:, ::-1, :].copy()
    if do_aug:                                                              # augment_image, nyu.py:229-245
        image_aug = image ** gamma
        image_aug = image_aug * brightness
        white = np.ones((image.shape[0], image.shape[1]))
        color_image = np.stack([white * colors[i] for i in range(3)], axis=2)
        image_aug *= color_image
        image = np.clip(image_aug, 0, 1)
    img = image.transpose(2, 0, 1)                                          # ToTensor.to_tensor, nyu.py:296-297
    if normalize:
        img = (img - MEAN[:, None, None]) / STD[:, None, None]              # transforms.Normalize (restated)
    return np.ascontiguousarray(img, dtype=np.float32), np.ascontiguousarray(depth.transpose(2, 0, 1), dtype=np.float32)


# ---- random rotation (nyu.py:121-124, 200-202) ----------------------------------------------------------------------------
# The reference calls PIL: `image.rotate(angle, resample=BILINEAR)` on the RGB image and `depth.rotate(angle, resample=NEAREST)`
# on the 16-bit depth png (mode "I;16").  Pillow is a third-party dependency of the reference (requirements.txt), not under
# /root/reference; it IS installed in this image (12.2.0), so this restatement of its published algorithm (Image.rotate ->
# Image.transform(AFFINE) -> libImaging/Geometry.c: ImagingGenericTransform + affine_transform + bilinear_filter32RGB /
# nearest_filter16) is pinned against Pillow itself (tests/golden/augment_rotate.npz, oracle/gen_golden_augment.py).
def rotate_matrix(angle_deg: float, w: int, h: int):
    """Image.rotate: the destination->source affine matrix (a, b, c, d, e, f) in float64, exactly as Pillow computes it."""
    import math
    angle = angle_deg % 360.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = w / 2, h / 2
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]          # note: uses the ORIGINAL m[2] = 0 in Pillow too (tuple assignment)
    m[2] += cx
    m[5] += cy
    return m


def _rotate_is_identity_case(angle_deg: float, w: int, h: int):
    a = angle_deg % 360.0
    return a == 0 or a == 180 or (a in (90, 270) and w == h)


def rotate_rgb_bilinear(img_u8: np.ndarray, angle_deg: float) -> np.ndarray:
    """img [H, W, 3] uint8 -> rotated (same size, black outside), Pillow BILINEAR."""
    h, w, _ = img_u8.shape
    assert not _rotate_is_identity_case(angle_deg, w, h), "Pillow takes a transpose fast path for multiples of 90 degrees"
    m = rotate_matrix(angle_deg, w, h)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    xin = m[0] * (xs + 0.5) + m[1] * (ys + 0.5) + m[2]                  # affine_transform
    yin = m[3] * (xs + 0.5) + m[4] * (ys + 0.5) + m[5]
    inside = (xin >= 0.0) & (xin < w) & (yin >= 0.0) & (yin < h)         # BILINEAR_HEAD
    xf, yf = xin - 0.5, yin - 0.5
    x = np.floor(xf).astype(np.int64)
    y = np.floor(yf).astype(np.int64)
    dx, dy = xf - x, yf - y
    x0, x1 = np.clip(x, 0, w - 1), np.clip(x + 1, 0, w - 1)
    y0 = np.clip(y, 0, h - 1)
    has_y1 = (y + 1 >= 0) & (y + 1 < h)
    y1 = np.clip(y + 1, 0, h - 1)
    src = img_u8.astype(np.float64)
    out = np.zeros_like(img_u8)
    for b in range(3):
        p = src[:, :, b]
        v1 = p[y0, x0] + (p[y0, x1] - p[y0, x0]) * dx
        v2 = np.where(has_y1, p[y1, x0] + (p[y1, x1] - p[y1, x0]) * dx, v1)
        v = v1 + (v2 - v1) * dy
        out[:, :, b] = np.where(inside, v, 0.0).astype(np.uint8)           # (UINT8) v1: truncation
    return out


def rotate_u16_nearest(dep_u16: np.ndarray, angle_deg: float) -> np.ndarray:
    """dep [H, W] uint16 (mode I;16) -> rotated, Pillow NEAREST (generic transform, double arithmetic)."""
    h, w = dep_u16.shape
    assert not _rotate_is_identity_case(angle_deg, w, h)
    m = rotate_matrix(angle_deg, w, h)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    xin = m[0] * (xs + 0.5) + m[1] * (ys + 0.5) + m[2]
    yin = m[3] * (xs + 0.5) + m[4] * (ys + 0.5) + m[5]
    x = np.where(xin < 0.0, -1, np.trunc(np.where(xin < 0.0, 0.0, xin))).astype(np.int64)     # COORD
    y = np.where(yin < 0.0, -1, np.trunc(np.where(yin < 0.0, 0.0, yin))).astype(np.int64)
    ok = (x >= 0) & (x < w) & (y >= 0) & (y < h)
    return np.where(ok, dep_u16[np.clip(y, 0, h - 1), np.clip(x, 0, w - 1)], 0).astype(np.uint16)
