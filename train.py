#!/usr/bin/env python3
"""Training with the reference's CLI and loop (`/root/reference/train.py:41-209`):

    python train.py @configs/cfpnet_combine1.txt [--synthetic N] [--max_steps K] [--stop_after K] [--seed S] [--save weights/x.pt] [--dtype bf16|f16|f32] [--eager] [--validate N]
                    [--resume checkpoints/x.pt] [--weight_path weights/x.pt] [--backend nccl|gloo] [--local_gpu I]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py @configs/cfpnet_combine1.txt --synthetic 4096

Per step: ToF simulation of the batch from its ground-truth depth (GPU), forward in training mode, SILog loss, backward of
the whole network (HIP tape, `cfpnet_amd/train_model.py`), gradient all-reduce over the data-parallel ranks (RCCL, flat
buffer), AdamW with OneCycle (lr and beta1 cycled like `train.py:82-94`, `clip_grad_norm_(0.1)` unless `--disable_clip_grad`).
One process per GPU with the global batch `--bs` split over the ranks (the reference uses nn.DataParallel on one process).

Without `--synthetic` the train split of `--filenames_file` under `--data_path` is read like the reference's loader does
(`cfpnet_amd.data.NYUTrainFiles`: PIL decode + border crop in worker threads, one batch ahead).  Differences on purpose:
`--synthetic N` trains on N seeded synthetic samples per epoch (the NYU files are not on this box; without it a missing
`filenames_file` is an error) -- generated like the loader's PIL images (uint8 RGB and 16-bit depth at
456x608, the frame after the Kinect-border crop of nyu.py:117-118) and put through the loader's augmentation ON THE DEVICE:
random rotation when `--do_random_rotate` (Pillow-exact), random crop to the input size, flip, gamma / brightness / colour
jitter, normalisation (`cfpnet_amd/augment.py`; `--no_augment` feeds pre-cropped tensors instead); wandb logging is left out; validation (`--validate N`: N synthetic eval samples through the inference engine and the device-side
metrics) runs at the end of every epoch whose step count is a multiple of `--validate_every`, as in train.py:137-156, writing
`<epoch>_<rmse>.pt` and `best.pt` next to `--save`, and once more at the end.
bf16 activations with float32 master parameters by default (`--dtype`); the step is replayed as one HIP graph unless
`--eager`.  There is no PyTorch autograd or fallback anywhere in the step.
"""
import os
import sys
import time

import numpy as np
import torch


def _pop(argv, flag, default=None, cast=str):
    if flag in argv:
        i = argv.index(flag)
        v = cast(argv[i + 1])
        del argv[i:i + 2]
        return v
    return default


class SyntheticTrainSet:
    """Seeded stand-in for the NYU train split at the training crop size: random RGB, planes-and-boxes depth with holes."""

    def __init__(self, n, H, W, seed):
        self.n, self.H, self.W, self.seed = n, H, W, seed

    def raw_batch(self, index, bs, H0=456, W0=608):
        """What the NYU loader holds after opening + border-cropping the files (nyu.py:105-118): uint8 RGB [bs,H0,W0,3] and
        16-bit depth in millimetres [bs,H0,W0]."""
        from cfpnet_amd import synthetic
        imgs, deps = [], []
        for j in range(bs):
            i = (index * bs + j) % self.n
            rng = np.random.default_rng(self.seed + i)
            imgs.append(rng.integers(0, 256, (H0, W0, 3), dtype=np.uint8))
            d = synthetic.make_depth(H0, W0, seed=self.seed + 7919 * (i + 1), holes=0.1 * (i % 3))
            deps.append(np.clip(np.rint(d * 1000.0), 0, 65535).astype(np.uint16))
        return torch.from_numpy(np.stack(imgs)), torch.from_numpy(np.stack(deps).view(np.int16))

    def batch(self, index, bs):
        from cfpnet_amd import data, synthetic
        imgs, deps = [], []
        for j in range(bs):
            i = (index * bs + j) % self.n
            rng = np.random.default_rng(self.seed + i)
            rgb = rng.random((3, self.H, self.W), dtype=np.float32)
            imgs.append((rgb - data.IMAGENET_MEAN[:, None, None]) / data.IMAGENET_STD[:, None, None])
            deps.append(synthetic.make_depth(self.H, self.W, seed=self.seed + 7919 * (i + 1), holes=0.1 * (i % 3))[None])
        return torch.from_numpy(np.stack(imgs)), torch.from_numpy(np.stack(deps))


def _prefetch(make, n, depth=3):
    """make(i) for i in range(n) produced by a background thread, `depth` items ahead of the consumer (the synthetic samples are
    numpy work on the host; the training step itself is ~37 ms)."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)

    def run():
        try:
            for i in range(n):
                q.put(make(i))
            q.put(None)
        except BaseException as e:
            q.put(e)
    threading.Thread(target=run, daemon=True).start()
    while True:
        item = q.get()
        if item is None:
            return
        if isinstance(item, BaseException):
            raise item
        yield item


SHUFFLE_SEED = 117010053       # train.py:218 seeds everything with this number


def load_model_file(path, manifest_sd):
    """A `save_weights` file (bare state_dict) or a `save_checkpoint` file ({"model", "optimizer", "epoch"}), DataParallel's
    "module." prefix stripped (model_io.py:47-52), checked STRICTLY against the model's keys and shapes like
    `load_state_dict` does (model_io.py:16,54).  -> (state_dict, optimizer entry or None, epoch or None); for a checkpoint of
    this loop `epoch` is (last fully completed epoch, global step)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    opt, epoch = None, None
    if isinstance(ckpt, dict) and "model" in ckpt and not torch.is_tensor(ckpt["model"]):
        opt, epoch, gstep, ckpt = ckpt.get("optimizer"), ckpt.get("epoch"), ckpt.get("global_step"), ckpt["model"]
        if gstep is not None:                    # written by THIS loop: the real progress, also of a run cut short by --max_steps
            epoch = (epoch, int(gstep))
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ckpt.items()}
    want = {k: tuple(v.shape) for k, v in manifest_sd.items() if torch.is_tensor(v)}
    missing, unexpected = sorted(set(want) - set(sd)), sorted(set(sd) - set(want))
    bad = [k for k in want if k in sd and tuple(sd[k].shape) != want[k]]
    if missing or unexpected or bad:
        raise RuntimeError(f"{path}: state_dict does not match the model (missing {missing[:3]}{'...' if len(missing) > 3 else ''}, "
                           f"unexpected {unexpected[:3]}{'...' if len(unexpected) > 3 else ''}, shape mismatch {bad[:3]})")
    return sd, opt, epoch


def save_training_checkpoint(path, weights_now, trainer, epoch, global_step, data_rng=None):
    """`save_checkpoint` of model_io.py:25-31: {"model", "optimizer", "epoch"}; the optimizer entry is the flat AdamW state.
    `epoch` is the last FULLY completed epoch (-1: none yet) -- the reference's meaning, `--resume` continues at epoch + 1 -- and
    `global_step` the number of optimizer steps taken so far, so a run that stopped inside an epoch (--max_steps) resumes where it
    stopped instead of being taken for finished.  `data_rng`: the numpy Generator that draws the dropped zones (rank 0's): its state and
    that of the other generators of the loop go in, so a resumed single-process run repeats the uninterrupted one bit for bit
    (tests/test_train_step_gpu.py); the other ranks of a multi-process run restart their generators."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ck = {"model": weights_now, "optimizer": trainer.optimizer_state_dict(), "epoch": int(epoch), "global_step": int(global_step)}
    if data_rng is not None:
        # every generator the loop draws from: dropped zones (own Generator), crop / flip / jitter / rotation (`random`, `np.random`, as the
        # reference's loader), positional-encoding windows (torch's CPU generator)
        import random
        ck["data_rng"] = {"drop": data_rng.bit_generator.state, "python": random.getstate(), "numpy": np.random.get_state(),
                          "torch": torch.get_rng_state()}
    torch.save(ck, path)


def drop_zones(sim, s, drop, rng):
    """nyu.py:155-158 then :179, in the reference's ORDER: int(len * drop_hist) of each sample's valid zones (drawn WITH
    replacement) lose their validity BEFORE the sample points are computed, so a dropped zone enters the ToF encoder as an
    all-zero row (`fh = zeros; fh[mask] = ...`, dataloader.py:67) exactly like a zone without signal does at evaluation time."""
    m = s["mask"].cpu().numpy().copy()
    for b in range(m.shape[0]):
        idx = np.where(m[b])[0]
        if idx.size:
            m[b, rng.choice(idx, int(idx.size * drop))] = False
    mask = torch.from_numpy(m).to(s["mask"].device)
    return mask, sim.sample_points(s["fh"], mask)


def main(argv=None):
    from cfpnet_amd import config, geometry, spec, weights
    from cfpnet_amd.tof import TofSimulator, zone_layout
    from cfpnet_amd.trainer import Trainer

    argv = list(argv if argv is not None else sys.argv[1:])
    n_syn = _pop(argv, "--synthetic", 0, int)
    max_steps = _pop(argv, "--max_steps", 0, int)
    seed = _pop(argv, "--seed", None, int)               # seeds `random`, `np.random` and torch's CPU generator (the reference's loop is unseeded)
    if seed is not None:
        import random
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    stop_after = _pop(argv, "--stop_after", 0, int)      # leave after this many steps of THIS process without touching the schedule (preemption)
    save_path = _pop(argv, "--save", "", str)
    log_every = _pop(argv, "--log_every", 10, int)
    n_val = _pop(argv, "--validate", 0, int)
    dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[_pop(argv, "--dtype", "bf16")]
    backend = _pop(argv, "--backend", "nccl")        # "gloo": ranks may share one GPU (tests), buckets staged through the host
    local_gpu = _pop(argv, "--local_gpu", None, int)
    eager = "--eager" in argv
    no_augment = "--no_augment" in argv
    sync_loss = "--sync_loss" in argv                # global-batch SILog (three moments all-reduced), the reference's DataParallel semantics
    argv = [a for a in argv if a not in ("--eager", "--no_augment", "--sync_loss")]
    args = config.parse_args(argv) if argv else config.defaults()
    args.mode = "train"

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
        dist_mod.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_mod
    if local_gpu is not None:
        local = local_gpu
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # flags of the reference's CLI that this loop cannot honour are refused, not ignored
    if int(getattr(args, "train_zone_random_offset", 0) or 0) > 0:
        raise NotImplementedError("--train_zone_random_offset > 0 changes the zone geometry per step; the captured step has it fixed "
                                  "(the ToF simulator itself supports per-sample offsets: cfpnet_amd.tof.TofSimulator.simulate(offsets=...))")
    if float(getattr(args, "noise_prob", 0.0) or 0.0) > 1e-3 and rank == 0:
        print("note: --noise_prob has no effect in the reference either (nyu.py:159-163 adds the noise to a copy); nothing is added here", flush=True)

    H, W = int(args.input_height), int(args.input_width)
    per_rank = max(1, int(args.bs) // world)
    files = None
    if n_syn <= 0:
        from cfpnet_amd import data as data_mod
        fn = getattr(args, "filenames_file", None)
        if not fn or not os.path.exists(fn):
            raise FileNotFoundError(f"filenames_file '{fn}' is not on this box -- pass --synthetic N to train on synthetic samples")
        files = data_mod.NYUTrainFiles(args, rank, world)         # nyu.py:62-136: the real train split, decoded by worker threads
        n_syn = len(files)
    steps_per_epoch = max(1, n_syn // (per_rank * world))
    total_steps = int(args.epochs) * steps_per_epoch
    if max_steps:
        total_steps = min(total_steps, max_steps)
    layers = list(args.attention_layer)
    sd = weights.make_torch_state_dict(spec.model_manifest(layers, int(args.n_bins), int(args.zone_sample_num)))
    start_epoch, start_step, opt_state = 0, 0, None
    for path, is_resume in ((getattr(args, "weight_path", "") or "", False), (getattr(args, "resume", "") or "", True)):
        if not path:
            continue
        loaded, opt, ep = load_model_file(path, sd)                 # strict: same keys, same shapes (model_io.py:14-17,34-54)
        sd.update(loaded)
        if is_resume and ep is not None:
            # train.py:30-38 restores the weights only (its optimizer restore is commented out, :83-84); a checkpoint written by
            # THIS loop also carries the flat AdamW moments + step counter and the run continues where it stopped
            if isinstance(ep, tuple):                              # this loop's checkpoint: real progress
                start_step = min(int(ep[1]), total_steps)
                start_epoch = start_step // steps_per_epoch
            else:                                                  # the reference's: epoch = last finished epoch
                start_epoch = int(ep) + 1
                start_step = min(start_epoch * steps_per_epoch, total_steps)
            opt_state = opt if isinstance(opt, dict) and opt.get("format", "").startswith("cfpnet_amd.") else None
    tr = Trainer(sd, layers, lr=float(args.lr), total_steps=max(total_steps, 2), weight_decay=float(args.wd), div_factor=float(args.div_factor),
                 final_div_factor=float(args.final_div_factor), hist_encoder_10x=bool(args.hist_encoder_10x),
                 clip_grad_norm=None if args.disable_clip_grad else 0.1, device=dev, dist=dist, world=world, n_bins=int(args.n_bins),
                 min_val=float(args.min_depth), max_val=float(args.max_depth), change_embedding=bool(args.change_embedding), dtype=dtype, no_skip_inside=bool(getattr(args, "no_skip_inside", False)),
                 norm=str(args.norm), sync_loss=sync_loss)
    if opt_state is not None:
        tr.load_optimizer_state_dict(opt_state)
    elif start_step > 0:
        # a reference-format checkpoint has no AdamW state to restore (train.py:83-84 is commented out there too): the moments restart
        # at zero, but the OneCycle schedule continues at the resumed step (what the reference means by `last_epoch`), so the
        # logged step and the learning rate stay consistent and the final anneal still happens
        tr.opt.step_count = start_step
        if rank == 0:
            print(f"resume without optimizer state: AdamW moments restart, OneCycle continues at step {start_step}/{total_steps}", flush=True)
    if sync_loss and dist is not None and backend != "nccl" and not eager:
        eager = True            # the moments' all-reduce sits inside the captured region; only RCCL collectives can be captured
        if rank == 0:
            print("note: --sync_loss under the gloo backend runs the eager step (a captured step needs RCCL)", flush=True)
    sim = TofSimulator(args, dev)
    zn, zp, _, _ = zone_layout(args, H, W)
    rects = geometry.centered_zone_rects(H, W, zn, zp)
    pi = geometry.collate_patch_info([geometry.patch_info_from_rect_data(rects, (H, W))] * per_rank)
    patch_info = {s: {k: torch.from_numpy(v) for k, v in pi[s].items()} for s in (4, 8, 16)}
    patch_info["zone_num"] = torch.from_numpy(pi["zone_num"])
    ds = SyntheticTrainSet(n_syn, H, W, seed=1000 + rank)
    rng = np.random.default_rng(4242 + rank)
    resumed_rng = None
    if rank == 0 and getattr(args, "resume", ""):      # a checkpoint of this loop carries rank 0's generator states (other ranks restart theirs)
        st = torch.load(args.resume, map_location="cpu", weights_only=False)
        if isinstance(st, dict) and isinstance(st.get("data_rng"), dict) and "drop" in st["data_rng"]:
            resumed_rng = st["data_rng"]
        del st
    drop = float(args.drop_hist)
    best_rmse, last_validated = float("inf"), -1

    def validate(at_step, epoch=None):
        """train.py:137-156,163-199: eval-mode forward of the current weights (inference engine), metrics on the device; with
        `--save` the epoch's weights go next to it as `<epoch>_<rmse>.pt` and the best ones (by rmse) as `best.pt`."""
        nonlocal best_rmse, last_validated
        from cfpnet_amd import data, metrics
        from cfpnet_amd.engine import Engine
        torch.cuda.synchronize()
        weights_now = tr.state_dict()
        # the drop-in boundary's default numerics (float32 storage, f16x3 matrix math: inside the 1e-3 gate on every weight family)
        eng = Engine(weights_now, layer_names=layers, n_bins=int(args.n_bins), min_val=float(args.min_depth), max_val=float(args.max_depth),
                     device=dev)
        build = data.EvalInputBuilder(args, dev)
        avg = metrics.RunningAverageDict()
        for img, dep, _ in data.batches(data.SyntheticEvalSamples(n_val, 480, 640, seed=99), 8):
            inp, gt = build(img, dep)
            _, pred, _ = eng.forward(inp, return_prob=False)
            avg.update(metrics.eval_metrics(pred, gt, float(args.min_depth_eval), float(args.max_depth_eval), mode=metrics.VALIDATE))
        m = avg.get_value()
        print(f"Validation metrics (step {at_step}):", {k: round(v, 3) for k, v in m.items()}, flush=True)
        if save_path:
            d = os.path.dirname(os.path.abspath(save_path))
            os.makedirs(d, exist_ok=True)
            if epoch is not None:                                    # train.py:150-155: checkpoint {model, optimizer, epoch} + bare weights
                torch.save(weights_now, os.path.join(d, f"{epoch}_{m['rmse']:.3f}.pt"))
                if steps_here > 0:                                   # the optimizer state exists once a step ran in this process
                    # the `epoch` FIELD is the last fully completed epoch (a mid-epoch validation has not finished `epoch` yet): a
                    # reference-style consumer that resumes at epoch + 1 must not skip the rest of this one
                    save_training_checkpoint(os.path.join(d, f"checkpoint_{epoch}.pt"), weights_now, tr, at_step // steps_per_epoch - 1, at_step, rng)
            if m["rmse"] < best_rmse:
                torch.save(weights_now, os.path.join(d, "best.pt"))
        best_rmse = min(best_rmse, m["rmse"])
        last_validated = at_step
        del eng
        return m

    t0, seen, step, steps_here = time.perf_counter(), 0, start_step, 0
    loss = torch.zeros(())
    if resumed_rng is not None:          # set here: everything above (model init, simulator tables) has drawn what it draws in any run
        import random
        rng.bit_generator.state = resumed_rng["drop"]
        random.setstate(resumed_rng["python"]); np.random.set_state(resumed_rng["numpy"]); torch.set_rng_state(resumed_rng["torch"])
    if start_step >= total_steps and rank == 0:
        print(f"nothing left to train: the checkpoint is at step {start_step} of {total_steps}", flush=True)
    for epoch in range(start_epoch, int(args.epochs)):
        # one global permutation per epoch, the same on every rank (each takes its slice of every global batch)
        file_batches = files.epoch_batches(per_rank, generator=torch.Generator().manual_seed(SHUFFLE_SEED + epoch)) if files is not None else None
        if files is None and not no_augment:
            file_batches = _prefetch(lambda i, e=epoch: ds.raw_batch(e * steps_per_epoch + i, per_rank) + (None,), steps_per_epoch)
        for i in range(steps_per_epoch):
            if step >= total_steps or (stop_after and steps_here >= stop_after):
                break
            if epoch * steps_per_epoch + i < start_step:          # resumed inside this epoch: these batches were consumed before
                if file_batches is not None:
                    next(file_batches)
                continue
            if no_augment and files is None:
                img, dep = ds.batch(epoch * steps_per_epoch + i, per_rank)
                depd = dep.to(dev)
            else:                                             # nyu.py:120-136 on the device, draws on the host in the loader's order
                from cfpnet_amd import augment
                raw_rgb, raw_dep, _ = next(file_batches)
                raw_rgb, raw_dep = raw_rgb.to(dev), raw_dep.to(dev)
                angles, params = [], []
                for _ in range(per_rank):
                    angles.append(augment.draw_rotation(float(args.degree)) if args.do_random_rotate else 0.0)
                    params.append(augment.draw_params(raw_rgb.shape[1], raw_rgb.shape[2], H, W))
                if args.do_random_rotate:
                    raw_rgb, raw_dep = augment.rotate(raw_rgb, raw_dep, angles)
                img, depd = augment.augment(raw_rgb, raw_dep, params, H, W)
            s = sim.simulate(depd)
            mask, hist_data = s["mask"], s["hist_data"]
            if drop > 1e-3:
                mask, hist_data = drop_zones(sim, s, drop, rng)
            inp = {"rgb": img, "additional": {"hist_data": hist_data, "rect_data": s["rect_data"], "mask": mask, "patch_info": patch_info}}
            if not eager and tr._graph is None:
                tr.capture(inp, depd)                         # the whole step as one HIP graph from here on
            loss, lr, beta1 = tr.step(inp, depd)
            step += 1
            steps_here += 1
            seen += per_rank * world
            if rank == 0 and (step % log_every == 0 or step == 1 or step == total_steps):
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                print(f"epoch {epoch + 1} step {step}/{total_steps} loss {float(loss):.4f} lr {lr:.2e} beta1 {beta1:.3f} {seen / dt:.1f} samples/s", flush=True)
        if rank == 0 and n_val > 0 and step % max(1, int(args.validate_every)) == 0 and step != last_validated:      # train.py:137: end of epoch
            validate(step, epoch)
        if stop_after and steps_here >= stop_after:
            break
    torch.cuda.synchronize()
    if rank == 0 and n_val > 0 and last_validated != step:
        validate(step)
    if rank == 0 and save_path:
        os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
        torch.save(tr.state_dict(), save_path)        # the reference's `model.state_dict()` file (model_io.py:14-17)
        if steps_here > 0:            # the last FULLY completed epoch (-1: none) + the global step: a cut-short run resumes, not "finished"
            save_training_checkpoint(os.path.splitext(save_path)[0] + ".ckpt.pt", tr.state_dict(), tr, step // steps_per_epoch - 1, step, rng)
    if dist:
        dist.barrier(); dist.destroy_process_group()
    return float(loss)


if __name__ == "__main__":
    main()
