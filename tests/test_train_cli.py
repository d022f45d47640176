"""Host logic of train.py that the GPU is not needed for: strict checkpoint loading (model_io.py:14-17,34-54), the
drop_hist-before-sampling order of the reference's loader (nyu.py:155-158,179), the per-epoch shuffle shared by all ranks."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import train as train_cli  # noqa: E402
from oracle import tof_oracle as TO  # noqa: E402


def _sd():
    return {"a.weight": torch.randn(4, 3), "a.bias": torch.randn(4), "bn.num_batches_tracked": torch.tensor(0)}


def test_load_model_file_accepts_bare_and_checkpoint_and_module_prefix(tmp_path):
    sd = _sd()
    torch.save(sd, tmp_path / "w.pt")
    got, opt, ep = train_cli.load_model_file(str(tmp_path / "w.pt"), sd)
    assert opt is None and ep is None and all(torch.equal(got[k], sd[k]) for k in sd)
    torch.save({"model": {"module." + k: v for k, v in sd.items()}, "optimizer": {"format": "cfpnet_amd.FlatAdamW/1"}, "epoch": 3}, tmp_path / "c.pt")
    got, opt, ep = train_cli.load_model_file(str(tmp_path / "c.pt"), sd)
    assert ep == 3 and opt["format"].startswith("cfpnet_amd.") and set(got) == set(sd)


@pytest.mark.parametrize("mutate", ["missing", "unexpected", "shape"])
def test_load_model_file_is_strict(tmp_path, mutate):
    sd = _sd()
    bad = dict(sd)
    if mutate == "missing":
        del bad["a.bias"]
    elif mutate == "unexpected":
        bad["module.extra"] = torch.zeros(1)
    else:
        bad["a.weight"] = torch.zeros(4, 5)
    torch.save(bad, tmp_path / "b.pt")
    with pytest.raises(RuntimeError, match="does not match the model"):
        train_cli.load_model_file(str(tmp_path / "b.pt"), sd)


class _HostSim:
    """sample_points of the simulator, restated on the host by the oracle (uniform branch)."""

    def __init__(self):
        self.w = TO.linspace_weights_f32(16)

    def sample_points(self, fh, mask):
        out = [TO.sample_points_uniform(fh[b].numpy(), mask[b].numpy(), *self.w) for b in range(fh.shape[0])]
        return torch.from_numpy(np.stack(out))


def test_drop_hist_is_applied_before_the_sample_points_like_the_loader():
    """nyu.py:155-158 drops zones from the mask BEFORE sample_point_from_hist_parallel (:179): a dropped zone's 16 samples are
    zeros, the kept ones are unchanged."""
    rng = np.random.default_rng(0)
    B, Z = 3, 36
    fh = torch.from_numpy(np.stack([rng.uniform(0.5, 3.0, (B, Z)), rng.uniform(0.02, 0.2, (B, Z))], -1))
    mask = torch.ones(B, Z, dtype=torch.bool)
    mask[0, :5] = False
    sim = _HostSim()
    s = {"mask": mask, "fh": fh, "hist_data": sim.sample_points(fh, mask)}
    m2, h2 = train_cli.drop_zones(sim, s, 0.34, np.random.default_rng(4))
    dropped = mask & ~m2
    assert 0 < int(dropped.sum()) <= B * int(Z * 0.34) and not (m2 & ~mask).any()
    assert not h2[dropped].any() and s["hist_data"][dropped].abs().sum() > 0          # the samples went with the validity
    assert torch.equal(h2[m2], s["hist_data"][m2])
    for b in range(B):                                                               # int(len * drop) draws WITH replacement
        assert int(dropped[b].sum()) <= int(int(mask[b].sum()) * 0.34)


def test_epoch_shuffle_is_shared_by_the_ranks(tmp_path):
    import json
    from cfpnet_amd import data
    names = [{"filename": f"x/scene/rgb_{i:05d}.jpg"} for i in range(40)]
    fn = tmp_path / "split.json"
    fn.write_text(json.dumps({"train": names}))
    args = types.SimpleNamespace(filenames_file=str(fn), data_path=str(tmp_path), num_threads=1)
    seen = []
    for rank in range(2):
        ds = data.NYUTrainFiles(args, rank, 2)
        ds._batch = lambda idx: idx                                                   # no files on this box: keep the indices
        seen.append([i for chunk in ds.epoch_batches(4, generator=torch.Generator().manual_seed(train_cli.SHUFFLE_SEED + 1)) for i in chunk])
    assert len(seen[0]) == len(seen[1]) == 20 and not set(seen[0]) & set(seen[1])     # disjoint halves of ONE permutation
    assert sorted(seen[0] + seen[1]) == list(range(40))
    with pytest.raises(ValueError, match="seeded identically"):
        next(iter(data.NYUTrainFiles(args, 0, 2).epoch_batches(4)))
