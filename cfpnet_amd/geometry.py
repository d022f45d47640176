"""Integer zone geometry of the ToF sensor (host side, bit-exact with the reference).

Rows G1/G2 of SURVEY.md §8(a):
  * `patch_info_from_rect_data`  <- `/root/reference/src/utils/dataloader.py:13-40`
  * `sample_points_from_hist`    <- `/root/reference/src/utils/dataloader.py:65-80` (uniform branch)
  * `FusionGeometry.from_patch_info` <- `/root/reference/src/models/fusion.py:67-84,103-120`
  * `lsa_padding`, `gsa_keys`    <- `/root/reference/src/models/transformer.py:100-105,132,146`

The reference materialises boolean masks (`zone_mask [B,HW,D]`, `hist_mask`, `pad_mask`); every
one of them is a clipped rectangle or a per-zone broadcast, so the product never builds them:
the HIP kernels receive the handful of integers computed here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np


def _as_np(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def _trunc_i32(v) -> int:
    # torch `.to(torch.int32)` on a float tensor truncates toward zero
    return int(np.trunc(np.float32(v)))


def patch_info_from_rect_data(rect_data, image_hw: Tuple[int, int] = (480, 640)) -> Dict:
    """rect_data [Z,4] float32 (sy, sx, ey, ex) in pixels -> per-scale integer geometry.

    `image_hw` defaults to the reference's hard-coded 480x640 (dataloader.py:21,23), which it
    uses even for 416x544 training crops.  Returns numpy int32 arrays shaped like the reference's
    (un-collated) tensors.
    """
    r = _as_np(rect_data).astype(np.float32)
    zone_num = int(math.sqrt(r.shape[0]))
    max_ph = _trunc_i32(np.max(r[..., 2] - r[..., 0]))
    max_pw = _trunc_i32(np.max(r[..., 3] - r[..., 1]))
    ih, iw = np.float32(image_hw[0]), np.float32(image_hw[1])
    pad_h_px = int(max(np.max(np.abs(np.minimum(r[..., 0], np.float32(0)))),
                       np.max(np.maximum(r[..., 2], ih) - ih)))
    pad_w_px = int(max(np.max(np.abs(np.minimum(r[..., 1], np.float32(0)))),
                       np.max(np.maximum(r[..., 3], iw) - iw)))
    ret: Dict = {}
    for s in (4, 8, 16):
        fs = np.float32(s)
        ret[s] = {
            "pad_size": np.array([math.ceil(pad_h_px / s), math.ceil(pad_w_px / s)], dtype=np.int32),
            "patch_size": np.array([math.ceil(max_ph / s), math.ceil(max_pw / s)], dtype=np.int32),
            "index_wo_pad": np.array([
                _trunc_i32(np.min(r[..., 0] / fs)), _trunc_i32(np.min(r[..., 1] / fs)),
                _trunc_i32(np.max(r[..., 2] / fs)), _trunc_i32(np.max(r[..., 3] / fs))], dtype=np.int32),
        }
    ret["zone_num"] = zone_num
    return ret


def collate_patch_info(infos) -> Dict:
    """What torch's default collate does to a list of per-sample patch_info dicts: stack a batch
    dimension onto every leaf (`train.py:112` keeps the result on the host)."""
    out: Dict = {}
    for s in (4, 8, 16):
        out[s] = {k: np.stack([np.asarray(i[s][k]) for i in infos]) for k in ("pad_size", "patch_size", "index_wo_pad")}
    out["zone_num"] = np.array([i["zone_num"] for i in infos], dtype=np.int64)
    return out


def sample_points_from_hist(mu_sigma, mask, zone_sample_num: int = 16) -> np.ndarray:
    """Uniform branch (`--sample_uniform`): zone (mu, sigma) -> `zone_sample_num` depths on
    [mu-3sigma, mu+3sigma]; invalid zones stay zero (dataloader.py:67,74-79).
    out = w_start*start + w_end*end with w = linspace(1,0,n) / linspace(0,1,n), in float32."""
    ms = _as_np(mu_sigma).astype(np.float32)
    mk = _as_np(mask).astype(bool)
    n = zone_sample_num
    sigma = ms[:, 1]
    start = ms[:, 0] - np.float32(3.0) * sigma
    end = ms[:, 0] + np.float32(3.0) * sigma
    import torch  # torch.linspace defines the float32 weights the reference uses
    w0 = torch.linspace(1, 0, steps=n).numpy()
    w1 = torch.linspace(0, 1, steps=n).numpy()
    out = w0[None, :] * start[:, None] + w1[None, :] * end[:, None]
    out = out.astype(np.float32)
    out[~mk] = 0
    return out


def centered_zone_rects(height: int, width: int, zone_num: int, zone_px: int, offset: int = 0) -> np.ndarray:
    """The zone grid `get_hist_parallel` lays over the image (dataloader.py:93-103,121-123):
    a centred zone_num x zone_num grid of zone_px squares, row-major, as float32 (sy,sx,ey,ex)."""
    sy0 = int((height - zone_px * zone_num) / 2) + offset
    sx0 = int((width - zone_px * zone_num) / 2) + offset
    rects = np.zeros((zone_num * zone_num, 4), dtype=np.float32)
    for zy in range(zone_num):
        for zx in range(zone_num):
            sy, sx = sy0 + zy * zone_px, sx0 + zx * zone_px
            rects[zy * zone_num + zx] = (sy, sx, sy + zone_px, sx + zone_px)
    return rects


def _get(info, key):
    """patch_info is keyed by int 4/8/16; the reference indexes it with the float 640/W
    (fusion.py:41,71), which works because hash(4.0) == hash(4)."""
    if key in info:
        return info[key]
    return info[int(key)]


@dataclass(frozen=True)
class FusionGeometry:
    """Batch-reduced zone geometry at one decoder scale (fusion.py:70-84)."""

    zone_num: int
    pad_h: int
    pad_w: int
    p1: int
    p2: int
    sy_wo: int   # zone rectangle in (un-padded) token coordinates, may overhang the image
    sx_wo: int
    ey_wo: int
    ex_wo: int
    interpolate: bool

    @property
    def tzh(self) -> int:
        return self.ey_wo - self.sy_wo

    @property
    def tzw(self) -> int:
        return self.ex_wo - self.sx_wo

    @property
    def grid_h(self) -> int:   # per-zone grid the attention runs on
        return self.zone_num * self.p1

    @property
    def grid_w(self) -> int:
        return self.zone_num * self.p2

    def clipped(self, H: int, W: int) -> Tuple[int, int, int, int]:
        """zone_mask rectangle (fusion.py:104): (y0, y1, x0, x1) inside the H x W token map."""
        c = lambda v, hi: max(0, min(int(v), hi))
        return c(self.sy_wo, H), c(self.ey_wo, H), c(self.sx_wo, W), c(self.ex_wo, W)

    @staticmethod
    def from_patch_info(patch_info, conv_patch_size) -> "FusionGeometry":
        info = _get(patch_info, conv_patch_size)
        zn = int(_as_np(patch_info["zone_num"]).reshape(-1)[0])
        pad = _as_np(info["pad_size"]).reshape(-1, 2)
        ps = _as_np(info["patch_size"]).reshape(-1, 2)
        idx = _as_np(info["index_wo_pad"]).reshape(-1, 4)
        pad_h, pad_w = int(pad[:, 0].max()), int(pad[:, 1].max())
        p1, p2 = int(ps[:, 0].max()), int(ps[:, 1].max())
        sy, sx = int(idx[:, 0].min()), int(idx[:, 1].min())
        ey, ex = int(idx[:, 2].max()), int(idx[:, 3].max())
        interp = (ey - sy) != p1 * zn or (ex - sx) != p2 * zn
        return FusionGeometry(zn, pad_h, pad_w, p1, p2, sy, sx, ey, ex, bool(interp))


def lsa_padding(H: int, W: int, ws: int) -> Tuple[int, int]:
    """(pad_bottom, pad_right) so H, W become multiples of ws (transformer.py:100-103)."""
    return (ws - H % ws) % ws, (ws - W % ws) % ws


def gsa_keys(H: int, W: int, ws: int) -> Tuple[int, int]:
    """Output size of the stride-ws, kernel-ws, no-padding conv (floors; transformer.py:132,146)."""
    return (H - ws) // ws + 1, (W - ws) // ws + 1
