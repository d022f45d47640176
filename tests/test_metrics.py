"""Device-side evaluation metrics: oracle vs the reference's `compute_errors` goldens (CPU),
HIP kernel vs both (GPU, through the C ABI).

Tolerance: the reference reduces in float32 (numpy pairwise sums over ~3e5 pixels); the kernel evaluates the same
float32 per-pixel terms and sums in float64 -> 2e-5 relative on every metric, and the valid-pixel count is exact."""
import json
import os

import numpy as np
import pytest
import torch

from cfpnet_amd import synthetic
from oracle import metrics_oracle as MO

from helpers import GOLDEN

CASES = json.load(open(os.path.join(GOLDEN, "eval_metrics.json")))
IDS = [c["name"] for c in CASES]
KEYS = ("a1", "a2", "a3", "abs_rel", "rmse", "log_10", "rmse_log", "silog", "sq_rel")
RTOL = 2e-5


def _pair(c):
    return synthetic.make_eval_pair(c["H"], c["W"], c["Hp"], c["Wp"], c["seed"], c["holes"], c["noise"])


def _close(got, want):
    for k in KEYS:
        assert abs(got[k] - want[k]) <= RTOL * max(abs(want[k]), 1e-3), (k, got[k], want[k])


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_oracle_matches_reference_golden(c):
    gt, pred = _pair(c)
    g, p = MO.protocol_evaluate_all(pred, gt, c["lo"], c["hi"])
    assert g.size == c["n_valid"]
    _close(MO.compute_errors(g, p), c["evaluate_all"])
    g, p = MO.protocol_validate(pred, gt, c["lo"], c["hi"])
    _close(MO.compute_errors(g, p), c["validate"])


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_kernel_matches_reference_golden(c):
    from cfpnet_amd import metrics
    gt, pred = _pair(c)
    g, p = torch.from_numpy(gt)[None, None].cuda(), torch.from_numpy(pred)[None, None].cuda()
    for mode, name in ((metrics.EVALUATE_ALL, "evaluate_all"), (metrics.VALIDATE, "validate")):
        row = metrics.eval_metrics(p, g, c["lo"], c["hi"], mode=mode)[0].cpu().tolist()
        assert row[9] == c["n_valid"]
        _close(dict(zip(KEYS, row[:9])), c[name])


@pytest.mark.gpu
def test_kernel_batch_running_average_and_empty_image():
    """A batch of 8 (one image without any valid pixel): rows equal the per-image oracle, the running average skips the
    empty image like evaluate_all.py:83, repeated launches are bit-identical (fixed summation order)."""
    from cfpnet_amd import metrics
    B = 8
    pairs = [synthetic.make_eval_pair(480, 640, 240, 320, 700 + i, 0.1 * (i % 4), 0.1 + 0.05 * i) for i in range(B)]
    gts = np.stack([p[0] for p in pairs])
    preds = np.stack([p[1] for p in pairs])
    gts[5] = 0.0
    g, p = torch.from_numpy(gts).cuda(), torch.from_numpy(preds).cuda()
    rows = metrics.eval_metrics(p, g, 1e-3, 10.0)
    again = metrics.eval_metrics(p, g, 1e-3, 10.0)
    assert torch.equal(rows[torch.arange(B) != 5], again[torch.arange(B) != 5])
    r = rows.cpu().numpy()
    assert r[5, 9] == 0 and np.isnan(r[5, :9]).all()
    want = []
    for i in range(B):
        if i == 5:
            continue
        gg, pp = MO.protocol_evaluate_all(preds[i], gts[i], 1e-3, 10.0)
        w = MO.compute_errors(gg, pp)
        _close(dict(zip(KEYS, r[i, :9])), w)
        assert r[i, 9] == gg.size
        want.append(w)
    avg = metrics.RunningAverageDict()
    avg.update(rows[:3])
    avg.update(rows[3:])
    got = avg.get_value()
    for k in KEYS:
        assert abs(got[k] - np.mean([w[k] for w in want])) <= RTOL * max(abs(np.mean([w[k] for w in want])), 1e-3)


@pytest.mark.gpu
def test_compute_errors_drop_in_and_errors():
    from cfpnet_amd import metrics
    rng = np.random.default_rng(5)
    gt = rng.uniform(0.5, 9.0, 100_003).astype(np.float32)
    pred = (gt * np.exp(rng.normal(0, 0.2, gt.shape))).astype(np.float32)
    got = metrics.compute_errors(torch.from_numpy(gt).cuda(), torch.from_numpy(pred).cuda())
    _close(got, MO.compute_errors(gt, pred))
    assert list(got) == list(KEYS)
    with pytest.raises(ValueError):
        metrics.compute_errors(torch.zeros(4).cuda(), torch.zeros(5).cuda())
    with pytest.raises(RuntimeError, match="empty depth range"):
        metrics.eval_metrics(torch.ones(1, 4, 4).cuda(), torch.ones(1, 4, 4).cuda(), 2.0, 1.0)
