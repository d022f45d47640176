"""Config loader (config.py:4-121) and state-dict manifest against values
captured from the reference."""
import json
import os

import pytest

from cfpnet_amd import config, spec, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def misc(golden_dir):
    return json.load(open(os.path.join(golden_dir, "misc.json")))


def test_defaults_match_reference(misc):
    ours = vars(config.build_parser().parse_args([]))
    assert ours == misc["config_defaults"]


def test_txt_files_parse_like_reference(misc, tmp_path):
    # re-create each reference .txt from its parsed values' source flags: the shipped combine1 file
    ref = misc["configs"]["train_deltar_change_embedding_no_clip_grad_hist_encoder_optimized_10x_combine1.txt"]
    ns = config.parse_args(["@" + os.path.join(ROOT, "configs", "cfpnet_combine1.txt")])
    for k in ("bs", "n_bins", "norm", "input_height", "input_width", "min_depth", "max_depth", "zone_sample_num",
              "attention_layer", "change_embedding", "hist_encoder_10x", "no_skip_inside", "train_zone_num",
              "sample_uniform", "drop_hist", "disable_clip_grad", "lr", "wd", "epochs"):
        assert getattr(ns, k) == ref[k], k
    assert ns.batch_size == 16 and ns.num_workers == 12 and ns.mode == "train"


def test_atfile_whitespace_split(tmp_path):
    p = tmp_path / "a.txt"
    p.write_text("--bs 4   --n_bins 100\n\n--attention_layer hist2image image\n--change_embedding\n")
    ns = config.parse_args(["@" + str(p)])
    assert ns.bs == 4 and ns.n_bins == 100 and ns.attention_layer == ["hist2image", "image"] and ns.change_embedding


def test_yaml_loads_with_merge_rule():
    ns = config.parse_args([os.path.join(ROOT, "configs", "debug.yaml")])
    assert ns.attention_layer == ["hist2image", "image", "hist2image", "image"]
    assert ns.n_bins == 256 and ns.bs == 1 and ns.batch_size == 1
    assert ns.lr == 0.0003 and ns.zone_type == "8x8"      # parser defaults fill the rest
    ns2 = config.parse_args(["@" + os.path.join(ROOT, "configs", "debug.yaml")])
    assert vars(ns2) == vars(ns)


def test_cli_flags():
    ns = config.parse_args(["--n-bins", "64", "--same_lr", "--validate-every", "3"])
    assert ns.n_bins == 64 and ns.same_lr and ns.validate_every == 3


@pytest.mark.parametrize("tag,layers", [("combine1", spec.COMBINE1_LAYERS), ("baseline", spec.BASELINE_LAYERS)])
def test_manifest_matches_reference_state_dict(golden_dir, tag, layers):
    ref = json.load(open(os.path.join(golden_dir, "manifest.json")))[tag]
    ours = {k: list(s) for k, s, _ in spec.model_manifest(layers) if not k.startswith("img_encoder.")}
    assert ours == ref


def test_param_counts():
    m = spec.model_manifest(spec.COMBINE1_LAYERS)
    enc = spec.param_count([e for e in m if e[0].startswith("img_encoder.")])
    assert enc == 12_464_842 or abs(enc - 12.46e6) < 0.02e6, enc
    dec = spec.param_count([e for e in m if e[0].startswith("decoder.")])
    assert dec == 8_954_656
    assert spec.param_count([e for e in m if e[0].startswith("hist_encoder.")]) == 55_296
    assert spec.param_count([e for e in m if e[0].startswith("depth_head.")]) == 328_576
    dead = spec.param_count([e for e in m if spec.is_dead_param(e[0])])
    assert dead == 388_864


def test_weights_are_reproducible_and_key_addressed():
    a = weights.make_tensor("decoder.conv3.weight", (128, 256, 1, 1), "conv_lin")
    b = weights.make_tensor("decoder.conv3.weight", (128, 256, 1, 1), "conv_lin")
    c = weights.make_tensor("decoder.conv2.weight", (128, 256, 1, 1), "conv_lin")
    assert (a == b).all() and not (a == c).all()
    assert a.tobytes()[:8].hex() == weights.make_tensor("decoder.conv3.weight", (128, 256, 1, 1), "conv_lin").tobytes()[:8].hex()


def test_base_resolution_generalises_the_hard_coded_tables():
    """decoder.py:82-94 at 480x640 is the default; configs[4] (640x960) scales tables, windows (fusion.py:28) and the
    sr-conv kernels, and nothing else in the state dict."""
    assert spec.fusion_table() == spec.FUSION == {"cross_atten3": (128, (30, 40), 7), "cross_atten2": (64, (60, 80), 15),
                                                 "cross_atten1": (32, (120, 160), 31)}
    big = spec.fusion_table((640, 960))
    assert [big[k][1] for k in ("cross_atten3", "cross_atten2", "cross_atten1")] == [(40, 60), (80, 120), (160, 240)]
    assert [spec.window_size(big[k][1]) for k in ("cross_atten3", "cross_atten2", "cross_atten1")] == [7, 10, 14]
    m0 = {k: s for k, s, _ in spec.model_manifest(spec.COMBINE1_LAYERS)}
    m1 = {k: s for k, s, _ in spec.model_manifest(spec.COMBINE1_LAYERS, base_resolution=(640, 960))}
    assert list(m0) == list(m1)
    changed = sorted(k for k in m0 if m0[k] != m1[k])
    assert changed and all(k.endswith(("positional_encodings", "gsa.sr.weight")) for k in changed)
    assert m1["decoder.cross_atten1.positional_encodings"] == (160 * 240, 32)
    with pytest.raises(AssertionError):
        spec.fusion_table((500, 640))


def test_oracle_runs_config5_shape():
    """640x960 with 16x16 zones of 40 px through the CPU oracle (the GPU test compares the HIP engine against this)."""
    import torch
    from cfpnet_amd import synthetic
    from oracle import cfpnet_oracle as O
    base = (640, 960)
    sd = weights.make_torch_state_dict(spec.model_manifest(spec.COMBINE1_LAYERS, base_resolution=base))
    inp = synthetic.make_inputs(1, 640, 960, 16, 40, seed=3, drop_hist=0.2, image_hw=base)
    assert inp["additional"]["hist_data"].shape == (1, 256, 16)
    assert inp["additional"]["patch_info"][16]["patch_size"][0].tolist() == [3, 3]        # ceil(40 / 16): interpolated zones
    edges, pred, prob = O.forward(sd, inp, layer_names=spec.COMBINE1_LAYERS, base_resolution=base)
    assert pred.shape == (1, 1, 320, 480) and prob.shape == (1, 256, 320, 480) and bool(torch.isfinite(pred).all())
    with pytest.raises(RuntimeError):        # the reference's own tables (480x640) cannot take this map: broadcast error, as in fusion.py:94
        O.forward(weights.make_torch_state_dict(spec.model_manifest(spec.COMBINE1_LAYERS)), inp, layer_names=spec.COMBINE1_LAYERS)
