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


def test_nyu_train_files_loader(tmp_path):
    """`NYUTrainFiles` (nyu.py:93-118): file naming, the Kinect-border crop, raw uint8 / 16-bit pixels, per-rank slices of a
    shuffled epoch -- on files written here in the dataset's layout."""
    import json
    import types
    from PIL import Image
    from cfpnet_amd import data
    rng = np.random.default_rng(0)
    root = tmp_path / "nyu" / "train"
    names = []
    for scene, num in (("kitchen_0001", "00012"), ("kitchen_0001", "00045"), ("office_0003", "00007"), ("office_0003", "00100"), ("bath_0002", "00001")):
        (root / scene).mkdir(parents=True, exist_ok=True)
        dep = rng.integers(0, 10000, (480, 640), dtype=np.uint16)
        rgb = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
        Image.fromarray(dep).save(root / scene / f"sync_depth_{num}.png")
        Image.fromarray(rgb, "RGB").save(root / scene / f"rgb_{num}.jpg", quality=95)
        names.append({"filename": f"train/{scene}/{num}.h5"})
    fn = tmp_path / "nyu.json"
    fn.write_text(json.dumps({"train": names, "test": []}))
    args = types.SimpleNamespace(filenames_file=str(fn), data_path=str(root), num_threads=2)
    ds = data.NYUTrainFiles(args)
    assert len(ds) == 5 and ds.paths(2) == (str(root / "office_0003" / "rgb_00007.jpg"), str(root / "office_0003" / "sync_depth_00007.png"))
    rgb, dep = ds.load(1)
    with Image.open(root / "kitchen_0001" / "sync_depth_00045.png") as dm:                       # the reference's own calls (nyu.py:117-118)
        want_d = np.array(dm.crop((16, 12, 640 - 16, 480 - 12)))
    with Image.open(root / "kitchen_0001" / "rgb_00045.jpg") as im:
        want_i = np.array(im.crop((16, 12, 640 - 16, 480 - 12)))
    assert rgb.shape == (456, 608, 3) and rgb.dtype == np.uint8 and np.array_equal(rgb, want_i)
    assert dep.shape == (456, 608) and dep.dtype == np.uint16 and np.array_equal(dep, want_d)
    seen = []
    for r in range(2):                                                                             # two ranks, batch 1 each: disjoint slices
        d = data.NYUTrainFiles(args, rank=r, world=2)
        got = list(d.epoch_batches(1, generator=torch.Generator().manual_seed(7)))
        assert len(got) == 2 and all(b[0].shape == (1, 456, 608, 3) and b[0].dtype == torch.uint8 and b[1].dtype == torch.int16 for b in got)
        seen += [b[2][0] for b in got]
    assert len(set(seen)) == 4
    bad = types.SimpleNamespace(filenames_file=str(fn), data_path=str(tmp_path / "nowhere"), num_threads=1)
    with pytest.raises(FileNotFoundError):
        list(data.NYUTrainFiles(bad).epoch_batches(1))


def test_nyu_eval_files_loader(tmp_path):
    """`NYUEvalFiles` (nyu.py:72-76,101-107, online_eval branch): file naming under data_path_eval, no border crop, ToTensor
    normalisation, depth in metres."""
    import json
    from PIL import Image
    rng = np.random.default_rng(3)
    root = tmp_path / "nyu" / "test" / "bathroom"
    root.mkdir(parents=True)
    dep = rng.integers(0, 10000, (480, 640), dtype=np.uint16)
    rgb = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    Image.fromarray(dep).save(root / "sync_depth_00045.png")
    Image.fromarray(rgb, "RGB").save(root / "rgb_00045.jpg", quality=95)
    fn = tmp_path / "nyu.json"
    fn.write_text(json.dumps({"test": [{"filename": "test/bathroom/00045.h5"}], "train": []}))
    args = types.SimpleNamespace(filenames_file_eval=str(fn), data_path_eval=str(tmp_path / "nyu" / "test"))
    items = list(data.NYUEvalFiles(args))
    assert len(items) == 1
    img, d, name = items[0]
    assert name == "test/bathroom/00045.h5" and img.shape == (3, 480, 640) and d.shape == (1, 480, 640)
    with Image.open(root / "rgb_00045.jpg") as im:
        want = (np.asarray(im, dtype=np.float32) / 255.0 - data.IMAGENET_MEAN) / data.IMAGENET_STD
    assert np.allclose(img.numpy(), want.transpose(2, 0, 1), atol=1e-6)
    assert np.array_equal(d.numpy()[0], dep.astype(np.float32) / 1000.0)
