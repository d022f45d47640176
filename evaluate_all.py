#!/usr/bin/env python3
"""Evaluation with the reference's CLI and protocol (`/root/reference/evaluate_all.py:44-167`):

    python evaluate_all.py @configs/cfpnet_combine1.txt --selected_epoch best [--synthetic 64] [--dtype f32x3|f32|f16|bf16] [--bs 8]

Per batch: ToF simulation from the ground-truth depth (GPU, `cfp_tof_hist_sim`), model forward (HIP engine), then
`np.clip` -> bilinear to full resolution -> `min_depth < gt < max_depth` mask -> the nine `compute_errors` metrics, all in
one device kernel (`cfp_eval_metrics`, mode 0); the per-image rows stay on the device and the running average is
formed once at the end -- the reference moves prediction and ground truth to the host for every image.
Prints `Metrics: {...}` rounded to 3 decimals and the comma-joined line, like `evaluate_all.py:88-90`.

Differences on purpose: the xlsx report (openpyxl) is not written; `--synthetic N` evaluates N seeded synthetic samples
when the dataset is not on the box (without it a missing `filenames_file_eval` is an error); weights are the
deterministic key-addressed set unless `weights/<name>/<selected_epoch>.pt` (the reference's location) exists or
`--weight_path` names a checkpoint in the reference's state_dict layout.  There is no PyTorch fallback for the model,
the ToF simulation or the metrics.
"""
import os
import sys
import time

import torch


def _pop(argv, flag, default=None, cast=str):
    if flag in argv:
        i = argv.index(flag)
        v = cast(argv[i + 1])
        del argv[i:i + 2]
        return v
    return default


def main(argv=None):
    from cfpnet_amd import config, data, metrics
    from cfpnet_amd.deltar import make_model
    from cfpnet_amd.model_io import load_weights

    argv = list(argv if argv is not None else sys.argv[1:])
    n_syn = _pop(argv, "--synthetic", 0, int)
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32x3": "f32x3"}[_pop(argv, "--dtype", "f32x3")]
    bs = _pop(argv, "--batch", 8, int)
    args = config.parse_args(argv) if argv else config.defaults()
    device = torch.device("cuda:0")
    if n_syn > 0:
        samples = data.SyntheticEvalSamples(n_syn, 480, 640)
    else:
        fn = getattr(args, "filenames_file_eval", None)
        if not fn or not os.path.exists(fn):
            raise FileNotFoundError(f"filenames_file_eval '{fn}' not found -- pass --synthetic N to evaluate synthetic samples")
        samples = data.NYUEvalFiles(args)

    model = make_model(args, dtype=dtype)
    wp = getattr(args, "weight_path", "") or ""
    if not wp and str(getattr(args, "selected_epoch", "-1")) != "-1":
        cand = os.path.join("weights", str(args.name), f"{args.selected_epoch}.pt")
        wp = cand if os.path.exists(cand) else ""
    if wp:
        model = load_weights(model, wp)
    model = model.to(device).eval()
    build = data.EvalInputBuilder(args, device)
    avg = metrics.RunningAverageDict()
    n_img, t0 = 0, time.perf_counter()
    with torch.no_grad():
        for img, dep, names in data.batches(samples, bs):
            inp, gt = build(img, dep)
            _, pred, _, _ = model(inp)
            avg.update(metrics.eval_metrics(pred, gt, float(args.min_depth), float(args.max_depth), mode=metrics.EVALUATE_ALL))
            n_img += img.shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {k: round(v, 3) for k, v in avg.get_value().items()}
    print(f"Metrics: {res}")
    print(",".join(str(v) for v in res.values()))
    print(f"{n_img} images in {dt:.2f} s ({n_img / dt:.1f} images/s incl. host-side sample generation/decoding)", file=sys.stderr)
    return res


if __name__ == "__main__":
    main()
