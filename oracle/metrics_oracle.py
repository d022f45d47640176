"""CPU oracle for the evaluation metrics -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of `compute_errors` (`/root/reference/src/utils/metrics.py:4-24`) and of the two protocols around
it (`evaluate_all.py:38-41,80-84`; `train.py:187-199`).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this module.

Parity status: PINNED against the reference's own `compute_errors` (imported by `oracle/gen_golden_metrics.py` in the
build container; fixtures `tests/golden/eval_metrics.json`, checked by `tests/test_metrics.py`).
Tolerance: float32 reductions in the reference (numpy pairwise sums) -> 2e-5 relative.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def compute_errors(gt: np.ndarray, pred: np.ndarray) -> Dict[str, float]:
    """metrics.py:4-24 on two masked float32 vectors."""
    thresh = np.maximum(gt / pred, pred / gt)
    out = dict(a1=(thresh < 1.25).mean(), a2=(thresh < 1.25 ** 2).mean(), a3=(thresh < 1.25 ** 3).mean())
    out["abs_rel"] = np.mean(np.abs(gt - pred) / gt)
    out["rmse"] = np.sqrt(((gt - pred) ** 2).mean())
    out["log_10"] = np.abs(np.log10(gt) - np.log10(pred)).mean()
    out["rmse_log"] = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    err = np.log(pred) - np.log(gt)
    out["silog"] = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
    out["sq_rel"] = np.mean(((gt - pred) ** 2) / gt)
    return {k: float(v) for k, v in out.items()}


def protocol_evaluate_all(pred_lowres: np.ndarray, gt: np.ndarray, min_depth: float, max_depth: float):
    """evaluate_all.py:38-41,80-84: clip at model resolution, bilinear(align_corners) up, mask lo < gt < hi.
    pred_lowres [Hp,Wp], gt [H,W] float32 -> (gt[valid], pred[valid])."""
    p = np.clip(pred_lowres.astype(np.float32), min_depth, max_depth)
    p = F.interpolate(torch.from_numpy(p)[None, None], gt.shape[-2:], mode="bilinear", align_corners=True)[0, 0].numpy()
    valid = np.logical_and(gt > min_depth, gt < max_depth)
    return gt[valid], p[valid]


def protocol_validate(pred_lowres: np.ndarray, gt: np.ndarray, min_eval: float, max_eval: float):
    """train.py:187-199: bilinear up first, clamp / inf / nan fix-ups, mask with the *_eval bounds."""
    p = F.interpolate(torch.from_numpy(pred_lowres.astype(np.float32))[None, None], gt.shape[-2:], mode="bilinear",
                      align_corners=True)[0, 0].numpy()
    p[p < min_eval] = min_eval
    p[p > max_eval] = max_eval
    p[np.isinf(p)] = max_eval
    p[np.isnan(p)] = min_eval
    valid = np.logical_and(gt > min_eval, gt < max_eval)
    return gt[valid], p[valid]
