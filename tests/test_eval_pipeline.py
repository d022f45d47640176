"""End to end over the callers either side of the model: samples -> ToF simulation (GPU) -> forward (HIP engine) ->
clip/resize/mask/metrics (GPU) -> running average, through the reference's `evaluate_all.py` CLI, against the same chain
built from the three CPU oracles."""
import types

import numpy as np
import pytest
import torch

from cfpnet_amd import data, geometry, spec, synthetic, weights
from oracle import cfpnet_oracle as O
from oracle import metrics_oracle as MO
from oracle import tof_oracle as TO

KEYS = ("a1", "a2", "a3", "abs_rel", "rmse", "log_10", "rmse_log", "silog", "sq_rel")


def test_synthetic_samples_are_seeded_and_shaped():
    a = list(data.SyntheticEvalSamples(3, 96, 128))
    b = list(data.SyntheticEvalSamples(3, 96, 128))
    assert len(a) == 3 and all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, b))
    assert a[0][0].shape == (3, 96, 128) and a[0][1].shape == (1, 96, 128) and a[0][1].dtype == torch.float32
    got = list(data.batches(a, 2))
    assert [g[0].shape[0] for g in got] == [2, 1] and got[0][1].shape == (2, 1, 96, 128)


def _oracle_pipeline(n, bs, lo, hi):
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    rows = []
    for img, dep, _ in data.batches(data.SyntheticEvalSamples(n), bs):
        B, _, H, W = img.shape
        sims = [TO.get_hist(dep[b, 0].numpy()) for b in range(B)]
        infos = [geometry.patch_info_from_rect_data(s["fr"], (H, W)) for s in sims]
        pi = geometry.collate_patch_info(infos)
        pinfo = {s: {k: torch.from_numpy(v) for k, v in pi[s].items()} for s in (4, 8, 16)}
        pinfo["zone_num"] = torch.from_numpy(pi["zone_num"])
        inp = {"rgb": img, "additional": {"hist_data": torch.from_numpy(np.stack([s["pts"] for s in sims])),
                                          "rect_data": torch.from_numpy(np.stack([s["fr"] for s in sims])),
                                          "mask": torch.from_numpy(np.stack([s["mask"] for s in sims])), "patch_info": pinfo}}
        _, pred, _ = O.forward(sd, inp, layer_names=layers)
        for b in range(B):
            g, p = MO.protocol_evaluate_all(pred[b, 0].numpy(), dep[b, 0].numpy(), lo, hi)
            if g.size:
                rows.append(MO.compute_errors(g, p))
    return {k: float(np.mean([r[k] for r in rows])) for k in KEYS}


@pytest.mark.gpu
def test_evaluate_all_cli_matches_the_oracle_chain():
    import evaluate_all
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    got = evaluate_all.main(["@configs/cfpnet_combine1.txt", "--selected_epoch", "best", "--synthetic", "3", "--batch", "2", "--dtype", "f32"])
    want = _oracle_pipeline(3, 2, 1e-3, 10.0)
    assert list(got) == list(KEYS)
    for k in KEYS:
        assert abs(got[k] - want[k]) <= 2e-3 * max(abs(want[k]), 1e-2) + 6e-4, (k, got[k], want[k])     # printed values are rounded to 3 decimals


def test_missing_dataset_is_an_error_not_a_fallback():
    import evaluate_all
    with pytest.raises(FileNotFoundError):
        evaluate_all.main(["@configs/cfpnet_combine1.txt", "--selected_epoch", "best"])
