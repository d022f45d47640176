"""Command-line / config-file loader with the reference's flag surface.

Mirrors `/root/reference/src/config.py`: the same flag names, aliases, types and defaults
(config.py:14-93), `@file` expansion with whitespace-split lines (config.py:4-13), the
"single argument that names a .txt or .yaml file" dispatch (config.py:97-114) and the derived
fields `batch_size / num_threads / mode / num_workers` (config.py:118-121).

Differences, on purpose:
  * YAML works.  The reference re-parses a sibling `.txt` that does not exist for
    `configs/debug.yaml` (config.py:107) and exits; here YAML keys override parser defaults
    directly, which is the merge rule config.py:108-111 intended.
  * Importing this module never exits the interpreter: if `sys.argv` is not a CFPNet command
    line (pytest, a notebook), `args` holds the defaults.  Use `parse_args(argv)` explicitly.
"""
from __future__ import annotations

import argparse
import sys
from typing import List, Optional

# (flags, kwargs) -- one row per reference option
_S = "store_true"
_FLAGS = [
    (("--epochs",), dict(default=25, type=int)),
    (("--n-bins", "--n_bins"), dict(default=80, type=int)),
    (("--lr", "--learning-rate"), dict(default=0.0003, type=float)),
    (("--wd", "--weight-decay"), dict(default=0.1, type=float)),
    (("--div-factor", "--div_factor"), dict(default=25, type=float)),
    (("--final-div-factor", "--final_div_factor"), dict(default=100, type=float)),
    (("--bs",), dict(default=16, type=int)),
    (("--name",), dict(default="UnetAdaptiveBins")),
    (("--norm",), dict(default="linear", type=str, choices=["linear", "softmax", "sigmoid"])),
    (("--same-lr", "--same_lr"), dict(default=False, action=_S)),
    (("--resume",), dict(default="", type=str)),
    (("--notes",), dict(default="", type=str)),
    (("--tags",), dict(default="sweep", type=str)),
    (("--workers",), dict(default=11, type=int)),
    (("--dataset",), dict(default="nyu", type=str)),
    (("--dataset_eval",), dict(default="realsense", type=str)),
    (("--data_path",), dict(default="../dataset/nyu/sync/", type=str)),
    (("--filenames_file",), dict(default="./train_test_inputs/nyudepthv2_train_files_with_gt.txt", type=str)),
    (("--data_path_eval",), dict(default="../dataset/nyu/official_splits/test/", type=str)),
    (("--filenames_file_eval",), dict(default="./train_test_inputs/nyudepthv2_test_files_with_gt.txt", type=str)),
    (("--input_height",), dict(default=416, type=int)),
    (("--input_width",), dict(default=544, type=int)),
    (("--max_depth",), dict(default=10, type=float)),
    (("--min_depth",), dict(default=1e-3, type=float)),
    (("--do_random_rotate",), dict(default=False, action=_S)),
    (("--degree",), dict(default=2.5, type=float)),
    (("--min_depth_eval",), dict(default=1e-3, type=float)),
    (("--max_depth_eval",), dict(default=10, type=float)),
    (("--no_logging",), dict(action=_S)),
    (("--patch_size",), dict(default=16, type=int)),
    (("--zone_sample_num",), dict(default=16, type=int)),
    (("--save_for_demo",), dict(action=_S)),
    (("--save_rgb",), dict(action=_S)),
    (("--save_pred",), dict(action=_S)),
    (("--save_error_map",), dict(action=_S)),
    (("--save_entropy",), dict(action=_S)),
    (("--save_dir",), dict(default="tmp", type=str)),
    (("--weight_path",), dict()),
    (("--drop_hist",), dict(default=0.0, type=float)),
    (("--noise_mean",), dict(default=0.0, type=float)),
    (("--noise_sigma",), dict(default=0.0, type=float)),
    (("--noise_prob",), dict(default=0.0, type=float)),
    (("--train_zone_num",), dict(default=8, type=int)),
    (("--train_zone_random_offset",), dict(default=0, type=int)),
    (("--sample_uniform",), dict(action=_S)),
    (("--attention_layer",), dict(default=["hist2image", "image", "hist2image", "image"], nargs="+")),
    (("--validate-every", "--validate_every"), dict(default=100, type=int)),
    (("--simu_max_distance",), dict(default=4.0, type=float)),
    (("--model_name",), dict(default="deltar", type=str)),
    (("--d_type",), dict(default="uniform", type=str)),
    (("--random_simu_max_d",), dict(action=_S)),
    (("--simu_max_d",), dict(default=4.0, type=float)),
    (("--simu_min_d",), dict(default=3.0, type=float)),
    (("--use_my_cross",), dict(action=_S)),
    (("--test_refine",), dict(action=_S)),
    (("--save_residual",), dict(action=_S)),
    (("--save_residual_entropy",), dict(action=_S)),
    (("--save_gt",), dict(action=_S)),
    (("--change_embedding",), dict(action=_S)),
    (("--test_dataset",), dict(default="zjuL5", type=str)),
    (("--disable_clip_grad",), dict(action=_S)),
    (("--hist_encoder_10x",), dict(action=_S)),
    (("--no_skip_inside",), dict(action=_S)),
    (("--outside_zone_area_only",), dict(action=_S)),
    (("--zone_area_only",), dict(action=_S)),
    (("--zone_type",), dict(default="8x8", type=str)),
    (("--selected_epoch",), dict(default="-1", type=str)),
]


def _split_line(arg_line: str):
    for a in arg_line.split():
        if a.strip():
            yield str(a)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="CFPNet (MI355X-native) options", fromfile_prefix_chars="@",
                                conflict_handler="resolve")
    p.convert_arg_line_to_args = _split_line
    for flags, kw in _FLAGS:
        p.add_argument(*flags, **kw)
    return p


def _finish(ns: argparse.Namespace) -> argparse.Namespace:
    ns.batch_size = ns.bs
    ns.num_threads = ns.workers
    ns.mode = "train"
    ns.num_workers = ns.workers
    return ns


def load_yaml(path: str) -> argparse.Namespace:
    import yaml
    with open(path, "r") as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader) or {}
    ns = build_parser().parse_args([])
    merged = dict(vars(ns))
    merged.update(cfg)           # YAML keys win, parser defaults fill the rest (config.py:108-111)
    return _finish(argparse.Namespace(**merged))


def parse_args(argv: Optional[List[str]] = None) -> argparse.Namespace:
    """argv as after the program name.  A single `@file.txt` / `file.txt` / `[@]file.yaml`
    argument is treated as a config file like the reference does."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) == 1 and not argv[0].startswith("--"):
        a = argv[0]
        if a.endswith((".yaml", ".yml")) or "yaml" in a:
            return load_yaml(a.replace("@", ""))
        if "txt" in a:
            return _finish(build_parser().parse_args([a if a.startswith("@") else "@" + a]))
    return _finish(build_parser().parse_args(argv))


def defaults() -> argparse.Namespace:
    return _finish(build_parser().parse_args([]))


def _from_sys_argv() -> argparse.Namespace:
    try:
        return parse_args(None)
    except SystemExit:
        return defaults()
    except Exception:
        return defaults()


class _LazyArgs:
    """`from cfpnet_amd.config import args` -- resolved from sys.argv on first attribute access,
    so importing the package under pytest or inside another program has no side effects."""

    _ns = None

    def _resolve(self):
        if object.__getattribute__(self, "_ns") is None:
            object.__setattr__(self, "_ns", _from_sys_argv())
        return object.__getattribute__(self, "_ns")

    def __getattr__(self, k):
        return getattr(self._resolve(), k)

    def __setattr__(self, k, v):
        setattr(self._resolve(), k, v)

    def __repr__(self):
        return repr(self._resolve())


args = _LazyArgs()
