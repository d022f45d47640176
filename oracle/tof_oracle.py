"""CPU oracle for the ToF (L5 zone histogram) simulation -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of `get_hist_parallel` + both branches of `sample_point_from_hist_parallel`
(`/root/reference/src/utils/dataloader.py:83-134` and `:65-80`; call site `src/dataloader/nyu.py:154,179`).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Parity status: PINNED against outputs of the reference itself (`oracle/gen_golden_hist.py` imports
`src.utils.dataloader` in the build container; fixtures `tests/golden/hist_sim.npz`, checked by
`tests/test_tof_sim.py`).  Integer stages (histogram, cluster choice, mask, rectangles) are bit-exact; mu/sigma are
float64 sums whose order inside `torch.sum` is not part of the reference's contract, so they are pinned to 1e-12
relative and the float32 samples to 1 ulp.

What the reference computes per zone (one `zone_px` x `zone_px` patch of the ground-truth depth map):
  1. `torch.histc(patch, bins=int(max_d/0.04), min=0, max=max_d)` in float32.  ATen's CPU histc places an element
     at int((x-min)*bins/(max-min)) evaluated in float32 (HistogramKernel.cpp, linear interpolation WITHOUT the
     local edge search `torch.histogram` adds -- checked here against torch.histc on values within 3 ulp of every
     edge), ignores elements outside [min, max] and folds x == max into the last bin.
  2. bin 0 (< 4 cm: invalid/zero depth) is cleared, 20 counts of ambient floor are subtracted (clamped at 0).
  3. of the maximal runs of consecutive non-zero bins only the one with the largest sum survives (first on ties).
  4. n = sum(hist); mask = n > 0; with bin centres dist[i] = (float32(e[i+1]) + e[i]) / 2, e = arange(0, max_d+1e-9, 0.04)
     in float64:  mu = sum(dist*hist) / float32(n + 1e-9);  sigma = sqrt(sum(hist*(dist-mu)^2) / float32(n + 1e-9)) + 1e-9.
  5. samples: w0*(mu-3sigma) + w1*(mu+3sigma) in float64 with w0 = float32 linspace(1,0,S), w1 = float32 linspace(0,1,S),
     rounded to float32; invalid zones are zero.  Without `--sample_uniform` (the argparse default, dataloader.py:69-73):
     Normal(mu, sigma).icdf(ppf) = mu + sigma * erfinv(2 ppf - 1) * sqrt(2) at ppf = float32(arange(1e-3, 1, 0.998/(S-1))),
     where erfinv runs in float32 (ppf is a float32 tensor) and the products / sum in float64 (`sample_points_icdf`; pinned
     by `tests/golden/hist_sim_icdf.npz` = the reference's own output, `oracle/gen_golden_hist.py`).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

BIN_WIDTH = 0.04          # dataloader.py:93,100
AMBIENT_FLOOR = 20        # dataloader.py:106


def linspace_weights_f32(steps: int):
    """(w_start, w_end) of `tensor_linspace` (dataloader.py:42-57): float32 linspace(1,0,S) and linspace(0,1,S)
    evaluated ON THIS HOST.  ATen's CPU linspace is vectorised (base + lane*step per SIMD vector), so its last bit
    can differ between AVX2 and AVX-512 hosts; the two tables are therefore an INPUT of the restatement (and of the
    HIP kernel), produced the way the reference produces them.  Golden fixtures carry the generating host's tables."""
    import torch
    return torch.linspace(1, 0, steps).numpy().copy(), torch.linspace(0, 1, steps).numpy().copy()


def icdf_table_f32(steps: int) -> np.ndarray:
    """erfinv(2 ppf - 1) in float32 at the reference's ppf points (dataloader.py:70-71), evaluated ON THIS HOST by torch like
    the reference does; an input of the restatement for the same reason as the linspace tables."""
    import torch
    delta = 1e-3
    ppf = torch.Tensor(np.arange(delta, 1, (1 - 2 * delta) / (steps - 1)).tolist())
    return torch.erfinv(2 * ppf - 1).numpy().copy()


def sample_points_icdf(fh: np.ndarray, mask: np.ndarray, table: np.ndarray) -> np.ndarray:
    """Non-uniform branch (dataloader.py:69-73; torch.distributions.Normal.icdf: loc + scale * erfinv(2 v - 1) * sqrt(2)):
    fh [Z,2] float64 (mu, sigma), mask [Z] bool, table [S] float32 -> [Z,S] float32, zero where masked out.
    Order of operations as torch evaluates the expression: (sigma * t) * sqrt(2), then mu + that."""
    fh = np.asarray(fh, dtype=np.float64)
    t = np.asarray(table, dtype=np.float32).astype(np.float64)
    out = (fh[:, 0:1] + (fh[:, 1:2] * t[None, :]) * np.float64(math.sqrt(2))).astype(np.float32)
    out[~np.asarray(mask, dtype=bool)] = 0
    return out


def sample_points_uniform(fh: np.ndarray, mask: np.ndarray, w0: np.ndarray, w1: np.ndarray) -> np.ndarray:
    """Uniform branch (dataloader.py:74-79) on given (mu, sigma): the step-5 arithmetic of `get_hist`."""
    fh = np.asarray(fh, dtype=np.float64)
    start, end = fh[:, 0] - 3.0 * fh[:, 1], fh[:, 0] + 3.0 * fh[:, 1]
    out = (np.asarray(w0, np.float32).astype(np.float64)[None] * start[:, None]
           + np.asarray(w1, np.float32).astype(np.float64)[None] * end[:, None]).astype(np.float32)
    out[~np.asarray(mask, dtype=bool)] = 0
    return out


def zone_histogram(patch: np.ndarray, max_distance: float, bins: int) -> np.ndarray:
    """Step 1: int64 counts per bin of one patch (float32 values, any shape).  histc's CPU kernel (local_search off)
    places x at int(((x - min) * float(bins)) / (max - min)), every operation in float32."""
    x = np.asarray(patch, dtype=np.float32).ravel()
    lo, hi = np.float32(0.0), np.float32(max_distance)
    x = x[(x >= lo) & (x <= hi)]                                     # also drops NaN
    pos = (((x - lo) * np.float32(bins)) / (hi - lo)).astype(np.int64)
    pos[pos == bins] = bins - 1                                      # x == max belongs to the last bin
    return np.bincount(pos, minlength=bins).astype(np.int64)


def strongest_cluster(hist: np.ndarray) -> np.ndarray:
    """Steps 2-3 on one zone's counts (dataloader.py:105-114)."""
    h = hist.copy()
    h[0] = 0
    h = np.maximum(h - AMBIENT_FLOOR, 0)
    out = np.zeros_like(h)
    best_sum, best = -1, None
    i, nb = 0, len(h)
    while i < nb:
        if h[i] == 0:
            i += 1
            continue
        j = i
        while j < nb and h[j] != 0:
            j += 1
        s = int(h[i:j].sum())
        if s > best_sum:                                             # strict: first run wins ties (np.argmax)
            best_sum, best = s, (i, j)
        i = j
    if best is not None:
        out[best[0]:best[1]] = h[best[0]:best[1]]
    return out


def bin_centres(bins: int) -> np.ndarray:
    """dist of dataloader.py:116: a float32 tensor of upper edges plus a float64 array of lower edges, halved."""
    e = np.arange(bins + 1, dtype=np.float64) * BIN_WIDTH            # np.arange(0, max_d+1e-9, 0.04)
    return (e[1:].astype(np.float32).astype(np.float64) + e[:-1]) / 2.0


def get_hist(depth: np.ndarray, mode: str = "online_eval", train_zone_num: int = 8, max_distance: float = 4.0,
             offset: int = 0, zone_sample_num: int = 16, weights=None) -> Dict[str, np.ndarray]:
    """depth [H, W] float32 -> dict(hist [Z,bins] int64 (after cluster selection), fh [Z,2] f64, fr [Z,4] f32,
    mask [Z] bool, pts [Z,S] f32)."""
    depth = np.asarray(depth, dtype=np.float32)
    H, W = depth.shape
    bins = int(max_distance / BIN_WIDTH)
    zp = 64 if mode == "train" else 56
    zn = train_zone_num if mode == "train" else 8
    sy0 = int((H - zp * zn) / 2) + offset
    sx0 = int((W - zp * zn) / 2) + offset
    Z = zn * zn
    dist = bin_centres(bins)
    hist = np.zeros((Z, bins), dtype=np.int64)
    fh = np.zeros((Z, 2), dtype=np.float64)
    fr = np.zeros((Z, 4), dtype=np.float32)
    mask = np.zeros(Z, dtype=bool)
    pts = np.zeros((Z, zone_sample_num), dtype=np.float32)
    w0, w1 = weights if weights is not None else linspace_weights_f32(zone_sample_num)
    for zy in range(zn):
        for zx in range(zn):
            z = zy * zn + zx
            sy, sx = sy0 + zy * zp, sx0 + zx * zp
            fr[z] = (sy, sx, sy + zp, sx + zp)
            h = strongest_cluster(zone_histogram(depth[sy:sy + zp, sx:sx + zp], max_distance, bins))
            hist[z] = h
            n = int(h.sum())
            mask[z] = n > 0
            nf = np.float64(np.float32(np.float32(n) + np.float32(1e-9)))     # `n + 1e-9` stays float32
            acc = 0.0
            for i in np.nonzero(h)[0]:                                          # ascending-bin order
                acc += dist[i] * float(h[i])
            mu = acc / nf
            var = 0.0
            for i in np.nonzero(h)[0]:
                d = dist[i] - mu
                var += float(h[i]) * (d * d)
            sigma = np.sqrt(var / nf) + 1e-9
            fh[z] = (mu, sigma)
            if mask[z]:
                start, end = mu - 3.0 * sigma, mu + 3.0 * sigma
                pts[z] = (w0.astype(np.float64) * start + w1.astype(np.float64) * end).astype(np.float32)
    return dict(hist=hist, fh=fh, fr=fr, mask=mask, pts=pts)
