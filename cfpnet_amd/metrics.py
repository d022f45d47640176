"""Depth-evaluation metrics on the device: host-side mirror of the reference's evaluation helpers.

Reference interface: `compute_errors(gt, pred)` (`src/utils/metrics.py:4-24`), `RunningAverageDict`
(`src/utils/utils.py:14-41`) and the protocol around them in `evaluate_all.py:38-41,80-84` / `train.py:187-199`.
The arithmetic is the HIP kernel behind `cfp_eval_metrics` (`csrc/metrics.hip`); nothing here computes on the CPU.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import hip

KEYS = ("a1", "a2", "a3", "abs_rel", "rmse", "log_10", "rmse_log", "silog", "sq_rel")   # the reference's dict order
EVALUATE_ALL, VALIDATE = 0, 1


def eval_metrics(pred: torch.Tensor, gt: torch.Tensor, lo: float, hi: float, mode: int = EVALUATE_ALL,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pred [B,1,Hp,Wp] / [B,Hp,Wp] f32 at model resolution, gt [B,1,H,W] / [B,H,W] f32 -> [B,10] f64 on the device:
    the nine metrics in `KEYS` order plus the valid-pixel count.  No host synchronisation."""
    if pred.dim() == 4:
        pred = pred[:, 0]
    if gt.dim() == 4:
        gt = gt[:, 0]
    if pred.dtype != torch.float32 or gt.dtype != torch.float32 or not pred.is_cuda or not gt.is_cuda:
        raise ValueError("pred and gt must be float32 device tensors")
    pred, gt = pred.contiguous(), gt.contiguous()
    B, Hp, Wp = pred.shape
    if gt.shape[0] != B:
        raise ValueError("batch sizes differ")
    H, W = gt.shape[1:]
    # torch's bilinear kernel is not a copy at equal sizes: it reads each pixel twice with weights (1, 0), so in VALIDATE
    # order (interpolate, then clamp) an infinite prediction becomes NaN -> min_depth; that mode always interpolates
    interp = int((Hp, Wp) != (H, W) or mode == VALIDATE)
    nbytes = hip.load().cfp_eval_metrics_ws_bytes(B)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=pred.device)
    if out is None:
        out = torch.empty(B, 10, dtype=torch.float64, device=pred.device)
    hip.call("cfp_eval_metrics", pred.data_ptr(), Hp, Wp, gt.data_ptr(), H, W, B, interp, mode, lo, hi,
             ws.data_ptr(), nbytes, out.data_ptr(), hip.current_stream())
    return out


def compute_errors(gt: torch.Tensor, pred: torch.Tensor) -> Dict[str, float]:
    """Drop-in for `compute_errors(gt, pred)` on two already-masked float32 device vectors (any shape, same size)."""
    g, p = gt.reshape(1, 1, -1), pred.reshape(1, 1, -1)
    if g.numel() != p.numel() or g.numel() == 0:
        raise ValueError("gt and pred must be non-empty and the same size")
    r = eval_metrics(p, g, float("-inf"), float("inf"), mode=EVALUATE_ALL)[0].cpu().tolist()
    return dict(zip(KEYS, r[:9]))


class RunningAverageDict:
    """`RunningAverageDict` (src/utils/utils.py:27-41) over per-image metric rows kept on the device: `update` takes the
    [B,10] tensor of `eval_metrics`, skips images without valid pixels (evaluate_all.py:83) and applies the reference's
    running-mean recurrence avg <- (v + count*avg)/(count+1) image by image when the value is read."""

    def __init__(self):
        self._rows = []

    def update(self, rows: torch.Tensor) -> None:
        self._rows.append(rows)

    def get_value(self) -> Dict[str, float]:
        if not self._rows:
            return {}
        rows = torch.cat(self._rows, 0).cpu().tolist()            # the only host synchronisation
        avg, count = [0.0] * 9, 0
        for r in rows:
            if not r[9] > 0:
                continue
            avg = [(v + count * a) / (count + 1) for v, a in zip(r[:9], avg)]
            count += 1
        return dict(zip(KEYS, avg)) if count else {}
