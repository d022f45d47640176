"""The training step end to end: forward in training mode (batch-statistics BatchNorm), SILog loss and the backward of the
whole network on the HIP tape, against PyTorch autograd of the CPU oracle run the same way (`model.train()`,
`loss.backward()`, train.py:119-125).  Float32.  Tolerances: loss and prediction 1e-4 relative; every parameter gradient
relative to that tensor's largest reference entry (BatchNorm with batch statistics amplifies f32 summation-order noise,
most in the 40-layer encoder)."""
import numpy as np
import pytest
import torch

import os

from cfpnet_amd import spec, synthetic, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import cfpnet_oracle as O

pytestmark = pytest.mark.gpu


def _case(B=2, H=256, W=320, zn=3, zpx=64, seed=11):
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    inp = synthetic.make_inputs(B, H, W, zn, zpx, seed=seed, drop_hist=0.25)
    target = torch.from_numpy(np.stack([synthetic.make_depth(H, W, seed=seed + 5 + i, holes=0.1) for i in range(B)]))[:, None]
    offs = {"cross_atten3": (3, 5), "cross_atten2": (7, 2), "cross_atten1": (11, 30)}
    return layers, sd, inp, target, offs


def _oracle_step(layers, sd, inp, target, offs, dtype=torch.float32):
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    sdg = {}
    for k, v in sd.items():
        v = v.detach().clone()
        if v.is_floating_point():
            v = v.to(dtype)
            if not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        sdg[k] = v
    cast = lambda x: x.to(dtype) if torch.is_tensor(x) and x.is_floating_point() else x
    i2 = {"rgb": cast(inp["rgb"]), "additional": {k: cast(v) for k, v in inp["additional"].items()}}
    O.BN_TRAIN = True
    try:
        edges, pred, prob = O.forward(sdg, i2, layer_names=layers, pos_offsets=offs, grad=True)
        loss = O.silog_loss(pred, target.to(dtype), target > 1e-3, interpolate=True)
        loss.backward()
    finally:
        O.BN_TRAIN = False
    grads = {k: v.grad.double() for k, v in sdg.items() if getattr(v, "grad", None) is not None}
    stats = {k: v.detach().double() for k, v in sdg.items() if k.endswith(("running_mean", "running_var"))}
    return float(loss.detach()), pred.detach().double(), grads, stats


@pytest.mark.parametrize("geom", [dict(), dict(B=1, zn=4, zpx=56, seed=23)], ids=["zones64", "zones56_interp"])
def test_training_step_matches_autograd_of_the_oracle(geom):
    """Ground truth = float64 autograd of the oracle.  The float32 HIP step must be as close to it as PyTorch's own
    float32 autograd of the same model is (batch-statistics BatchNorm amplifies float32 summation-order noise to ~3e-3 of a
    tensor's largest gradient entry on this network; a conv bias in front of such a BatchNorm has an exactly-zero true
    gradient, so errors are measured against max(|g|_max, 1e-5 x the model's largest gradient entry))."""
    from cfpnet_amd.train_model import TrainNet
    layers, sd, inp, target, offs = _case(**geom)          # 56-px zones: 3.5-token zones at 1/16 -> the bilinear regroup path of hist2image
    loss64, pred64, g64, stats64 = _oracle_step(layers, sd, inp, target, offs, torch.float64)
    loss32, pred32, g32, _ = _oracle_step(layers, sd, inp, target, offs, torch.float32)
    net = TrainNet(sd, layers, "cuda:0")
    loss1, pred1, edges1 = net.forward_backward(inp, target, target > 1e-3, pos_offsets=offs)
    torch.cuda.synchronize()
    gh = {k: v.double().cpu() for k, v in net.grads().items()}
    print(f"loss f64 {loss64:.7f}  f32 oracle {loss32:.7f}  hip {float(loss1):.7f}")
    assert abs(float(loss1) - loss64) <= 2e-6 * abs(loss64) + abs(loss32 - loss64)
    assert float((pred1.double().cpu() - pred64).abs().max()) <= 1e-4 * float(pred64.abs().max())
    live = {k for k, g in g64.items() if float(g.abs().max()) > 0}
    assert not sorted(live - set(gh)), sorted(live - set(gh))[:10]                # every live parameter got a gradient
    assert not [k for k in gh if k not in g64]                                    # and no dead one did
    gmax = max(float(g.abs().max()) for g in g64.values())
    e_hip, e_ref = [], []
    for k in sorted(live):
        den = max(float(g64[k].abs().max()), 1e-5 * gmax)
        e_hip.append(float((gh[k] - g64[k]).abs().max()) / den)
        e_ref.append(float((g32[k] - g64[k]).abs().max()) / den)
    e_hip, e_ref = np.array(e_hip), np.array(e_ref)
    order = np.argsort(-e_hip)[:6]
    names = sorted(live)
    print(f"gradient error vs f64: median hip {np.median(e_hip):.2e} / f32 autograd {np.median(e_ref):.2e}; "
          f"worst hip {[(names[i], f'{e_hip[i]:.1e}', f'{e_ref[i]:.1e}') for i in order]}")
    assert np.median(e_hip) <= 1.5 * np.median(e_ref) + 1e-4
    assert np.quantile(e_hip, 0.99) <= 3 * np.quantile(e_ref, 0.99) + 1e-3
    assert e_hip.max() < 0.15
    for k, v in stats64.items():                                                  # running statistics: torch's momentum update
        assert torch.allclose(net.buf[k].double().cpu(), v, rtol=1e-3, atol=1e-5), k


def test_trainer_step_applies_adamw_to_its_gradients_and_loss_goes_down():
    """Plumbing of Trainer.step: gradients -> flat buffer -> AdamW(OneCycle) -> the parameters the next forward reads.
    The update must equal torch.optim.AdamW fed with the same gradients, and a few steps on one batch must reduce the loss."""
    from cfpnet_amd.trainer import Trainer
    from cfpnet_amd.train_model import TrainNet
    layers, sd, inp, target, offs = _case()
    tr = Trainer(sd, layers, lr=3e-4, total_steps=20, weight_decay=0.1)
    net = TrainNet(sd, layers, "cuda:0")
    net.forward_backward(inp, target, target > 1e-3, pos_offsets=offs)
    grads = {k: v.detach().cpu().clone() for k, v in net.grads().items()}
    loss0, lr, beta1 = tr.step(inp, target, pos_offsets=offs)
    torch.cuda.synchronize()
    sched = tr.opt.sched
    assert (lr, beta1) == sched.at(0) and abs(lr - 3e-4 / 25) < 1e-12
    ref = {k: torch.nn.Parameter(sd[k].detach().clone().float()) for k in grads}
    for k, p in ref.items():
        p.grad = grads[k].float()
    torch.optim.AdamW(ref.values(), lr=lr, betas=(beta1, 0.999), eps=1e-8, weight_decay=0.1).step()
    worst = max(float((tr.param(k).cpu() - p.detach()).abs().max()) / max(float(p.detach().abs().max()), 1e-3) for k, p in ref.items())
    assert worst < 1e-5, worst
    losses = [float(loss0)]
    for _ in range(6):
        l, _, _ = tr.step(inp, target, pos_offsets=offs)
        losses.append(float(l))
    print("losses", [round(x, 4) for x in losses])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    out = tr.state_dict()
    assert set(out) == set(sd) and all(tuple(out[k].shape) == tuple(sd[k].shape) for k in sd if torch.is_tensor(sd[k]))


def test_deltar_module_trains_through_torch_autograd():
    """The reference's own loop shape: `bin_edges, pred = model(input)` in train mode, a torch loss on `pred`,
    `loss.backward()` -- parameter .grad comes from the HIP tape through a torch.autograd.Function."""
    from cfpnet_amd.deltar import Deltar
    from cfpnet_amd.train_model import TrainNet
    import types
    layers, sd, inp, target, offs = _case()
    args = types.SimpleNamespace(attention_layer=layers, zone_sample_num=16, change_embedding=True, no_skip_inside=False, hist_encoder_10x=True)
    model = Deltar(n_bins=256, min_val=1e-3, max_val=10.0, norm="linear", args=args, dtype=torch.float32)
    model.load_state_dict(sd)
    model = model.to("cuda:0").train()
    rm_before = model.state_dict()["decoder.up1._net.1.running_mean"].clone()
    dinp = synthetic.to_device(inp, "cuda:0")
    edges, pred = model(dinp, pos_offsets=offs)
    assert pred.requires_grad and pred.shape == (2, 1, 128, 160) and edges.shape == (2, 257)
    tgt = target.to("cuda:0")
    loss = O.silog_loss(torch.clip(pred, 1e-3), tgt, tgt > 1e-3, interpolate=True)        # train.py:121-123 with torch ops on pred
    loss.backward()
    torch.cuda.synchronize()
    net = TrainNet(sd, layers, "cuda:0")
    loss1, _, _ = net.forward_backward(inp, target, target > 1e-3, pos_offsets=offs)
    ref = net.grads()
    assert abs(float(loss) - float(loss1)) < 1e-5 * float(loss1)
    named = dict(model.named_parameters())
    got_live = {k for k, p in named.items() if p.grad is not None}
    assert got_live == set(ref)
    gmax = max(float(g.abs().max()) for g in ref.values())
    errs = [float((named[k].grad - ref[k]).abs().max()) / max(float(ref[k].abs().max()), 1e-5 * gmax) for k in ref]
    # two float32 evaluations of the loss gradient apart (torch ops vs the SILog kernel), amplified by the BatchNorms
    assert np.median(errs) < 2e-3 and max(errs) < 0.15, (np.median(errs), max(errs))
    assert not torch.equal(model.state_dict()["decoder.up1._net.1.running_mean"], rm_before)      # running statistics moved
    with pytest.raises(RuntimeError):
        Deltar(n_bins=256, min_val=1e-3, max_val=10.0, norm="linear", args=args, dtype=torch.float32).train()(inp)


def test_deltar_module_graph_replay_equals_eager_tape():
    """`Deltar.train_graphs` (default): forward and backward of the module's training step replayed as two HIP graphs.  Three
    optimizer steps with NEW inputs and positional windows per step must leave exactly the parameters and running statistics
    of the same loop run launch by launch."""
    from cfpnet_amd.deltar import Deltar
    import types
    layers, sd, inp, target, offs = _case()
    args = types.SimpleNamespace(attention_layer=layers, zone_sample_num=16, change_embedding=True, no_skip_inside=False, hist_encoder_10x=True)
    batches = []
    for s_ in range(3):
        i2 = synthetic.make_inputs(2, 256, 320, 3, 64, seed=170 + s_, drop_hist=0.25 * (s_ % 2))
        t2 = torch.from_numpy(np.stack([synthetic.make_depth(256, 320, seed=190 + 2 * s_ + i, holes=0.1) for i in range(2)]))[:, None]
        o2 = {"cross_atten3": (s_, 2 * s_), "cross_atten2": (3 * s_, s_), "cross_atten1": (5 * s_, 7 * s_)}
        batches.append((synthetic.to_device(i2, "cuda:0"), t2.cuda(), o2))
    results = []
    for graphs in (False, True):
        model = Deltar(n_bins=256, min_val=1e-3, max_val=10.0, norm="linear", args=args, dtype=torch.float32)
        model.load_state_dict(sd)
        model = model.to("cuda:0").train()
        model.train_graphs = graphs
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.1)
        losses = []
        for dinp, tgt, o2 in batches:
            opt.zero_grad()
            edges, pred = model(dinp, pos_offsets=o2)
            loss = O.silog_loss(torch.clip(pred, 1e-3), tgt, tgt > 1e-3, interpolate=True)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        assert graphs == bool(model._train_captures)
        results.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (l0, s0), (l1, s1) = results
    assert l0 == l1, (l0, l1)
    # two forwards before one backward would read overwritten activations: refused loudly
    e1, p1 = model(batches[0][0], pos_offsets=batches[0][2])
    e2, p2 = model(batches[1][0], pos_offsets=batches[1][2])
    with pytest.raises(RuntimeError, match="overwritten"):
        p1.sum().backward()
    p2.sum().backward()
    assert all(torch.equal(s0[k], s1[k]) for k in s0), [k for k in s0 if not torch.equal(s0[k], s1[k])][:5]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_mixed_precision_training_step(dtype):
    """16-bit activations / matrix-core operands with float32 master parameters and parameter gradients: the loss stays within
    1 % of the float32 step, the gradient of every large tensor points the same way (cosine), and a few steps reduce the loss."""
    from cfpnet_amd.train_model import TrainNet
    from cfpnet_amd.trainer import Trainer
    layers, sd, inp, target, offs = _case()
    n32 = TrainNet(sd, layers, "cuda:0")
    l32, _, _ = n32.forward_backward(inp, target, target > 1e-3, pos_offsets=offs)
    g32 = n32.grads()
    n16 = TrainNet(sd, layers, "cuda:0", dtype=dtype)
    l16, _, _ = n16.forward_backward(inp, target, target > 1e-3, pos_offsets=offs)
    g16 = n16.grads()
    torch.cuda.synchronize()
    assert abs(float(l16) - float(l32)) < 1e-2 * float(l32)
    assert set(g16) == set(g32) and all(g.dtype == torch.float32 for g in g16.values())
    a = torch.cat([g16[k].reshape(-1) for k in sorted(g32)]).double()
    b = torch.cat([g32[k].reshape(-1) for k in sorted(g32)]).double()
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    print(f"{dtype}: loss {float(l16):.5f} vs f32 {float(l32):.5f}; cosine of the full gradient {cos:.4f}")
    assert cos > (0.97 if dtype == torch.float16 else 0.6)      # bf16: 8 significant bits through ~130 batch-statistics BatchNorms at batch 2
    tr = Trainer(sd, layers, lr=3e-4, total_steps=20, dtype=dtype)
    losses = [float(tr.step(inp, target, pos_offsets=offs)[0]) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_captured_training_step_equals_the_eager_one():
    """Trainer.capture(): the whole step (parameter re-layout, forward, loss, backward, gradient gathering) replayed as one
    HIP graph -- fed with NEW inputs and NEW positional-encoding windows at every step -- follows the eager trainer exactly."""
    from cfpnet_amd.trainer import Trainer
    layers, sd, inp, target, offs = _case()
    batches = []
    for s in range(3):
        i2 = synthetic.make_inputs(2, 256, 320, 3, 64, seed=70 + s, drop_hist=0.25 * (s % 2))
        t2 = torch.from_numpy(np.stack([synthetic.make_depth(256, 320, seed=90 + 2 * s + i, holes=0.1) for i in range(2)]))[:, None]
        o2 = {"cross_atten3": (s, 2 * s), "cross_atten2": (3 * s, s), "cross_atten1": (5 * s, 7 * s)}
        batches.append((synthetic.to_device(i2, "cuda:0"), t2.cuda(), o2))
    eager = Trainer(sd, layers, lr=3e-4, total_steps=20)
    graph = Trainer(sd, layers, lr=3e-4, total_steps=20)
    graph.capture(*batches[0][:2])
    for inp_b, tgt_b, offs_b in batches:
        l0, _, _ = eager.step(inp_b, tgt_b, pos_offsets=offs_b)
        l1, _, _ = graph.step(inp_b, tgt_b, pos_offsets=offs_b)
        torch.cuda.synchronize()
        assert abs(float(l0) - float(l1)) <= 1e-6 * abs(float(l0)), (float(l0), float(l1))
    assert torch.equal(eager.flat.param, graph.flat.param)            # same kernels, same order: bit-identical parameters


def test_training_step_variants_no_skip_inside_and_stale_embedding():
    """`--no_skip_inside` (fusion.py:156: the zone rectangle is replaced, not added to) and `change_embedding=False`
    (hist2image reads the first embedding): loss and gradients against float32 autograd of the oracle run the same way."""
    from cfpnet_amd.train_model import TrainNet
    layers, sd, inp, target, offs = _case()
    for kw in (dict(no_skip_inside=True, change_embedding=True), dict(no_skip_inside=False, change_embedding=False)):
        sdg = {}
        for k, v in sd.items():
            v = v.detach().clone()
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
            sdg[k] = v
        O.BN_TRAIN = True
        try:
            _, pred, _ = O.forward(sdg, inp, layer_names=layers, pos_offsets=offs, grad=True, **kw)
            loss0 = O.silog_loss(pred, target, target > 1e-3)
            loss0.backward()
        finally:
            O.BN_TRAIN = False
        net = TrainNet(sd, layers, "cuda:0", **kw)
        loss1, _, _ = net.forward_backward(inp, target, target > 1e-3, pos_offsets=offs)
        g1 = net.grads()
        assert abs(float(loss1) - float(loss0.detach())) < 1e-5 * float(loss0.detach()), kw
        g0 = {k: v.grad for k, v in sdg.items() if getattr(v, "grad", None) is not None and float(v.grad.abs().max()) > 0}
        assert set(g0) == set(g1), kw
        gmax = max(float(g.abs().max()) for g in g0.values())
        errs = [float((g1[k].cpu() - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-5 * gmax) for k in g0]
        assert np.median(errs) < 1e-2 and max(errs) < 0.2, (kw, np.median(errs), max(errs))
    with pytest.raises(NotImplementedError):
        TrainNet(sd, layers, "cuda:0", norm="softmax")


def test_train_cli_on_dataset_files(tmp_path):
    """train.py without --synthetic: files in the NYU layout -> PIL decode + border crop (worker threads) -> rotation, crop, flip,
    jitter on the device -> ToF simulation -> captured training step; saves a checkpoint the inference engine loads."""
    import json
    import sys
    from PIL import Image
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import train as train_cli
    from cfpnet_amd.engine import Engine
    rng = np.random.default_rng(1)
    root = tmp_path / "nyu" / "train" / "room_0001"
    root.mkdir(parents=True)
    names = []
    for i in range(4):
        dep = (synthetic.make_depth(480, 640, seed=300 + i, holes=0.05) * 1000).astype(np.uint16)
        Image.fromarray(dep).save(root / f"sync_depth_{i:05d}.png")
        Image.fromarray(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8), "RGB").save(root / f"rgb_{i:05d}.jpg")
        names.append({"filename": f"train/room_0001/{i:05d}.h5"})
    fn = tmp_path / "split.json"
    fn.write_text(json.dumps({"train": names}))
    out = tmp_path / "w" / "last.pt"
    loss = train_cli.main(["@" + os.path.join(ROOT, "configs", "cfpnet_combine1.txt"),
                           "--filenames_file", str(fn), "--data_path", str(tmp_path / "nyu" / "train"), "--bs", "2", "--epochs", "2",
                           "--do_random_rotate", "--save", str(out), "--log_every", "1"])
    assert np.isfinite(loss) and out.exists()
    sd = torch.load(out, map_location="cpu")
    layers = spec.COMBINE1_LAYERS
    eng = Engine(sd, layer_names=layers, dtype=torch.bfloat16)
    _, pred, _ = eng.forward(synthetic.to_device(synthetic.make_inputs(1, 480, 640, 8, 56, seed=3), "cuda:0"), return_prob=False)
    assert bool(torch.isfinite(pred).all())


def test_train_cli_stop_and_resume_repeats_the_uninterrupted_run(tmp_path):
    """Checkpoint / resume of train.py (model_io.py:25-31 + this loop's extras): a run that is stopped after 3 of its 8 steps
    (`--stop_after`, the schedule untouched) and resumed from the `.ckpt.pt` it left -- weights, running statistics, flat AdamW moments,
    OneCycle position, and the state of every generator the loop draws from (dropped zones, crop / flip / jitter / rotation, positional
    windows) -- ends with exactly the weights of the uninterrupted run."""
    import train as train_cli
    common = ["@" + os.path.join(ROOT, "configs", "cfpnet_combine1.txt"), "--synthetic", "8", "--bs", "2", "--epochs", "2", "--do_random_rotate",
              "--log_every", "100", "--seed", "7"]
    a, b1, b2 = tmp_path / "a" / "w.pt", tmp_path / "b" / "w.pt", tmp_path / "b2" / "w.pt"
    train_cli.main(common + ["--save", str(a)])
    train_cli.main(common + ["--save", str(b1), "--stop_after", "3"])
    ck = torch.load(str(b1)[:-3] + ".ckpt.pt", map_location="cpu", weights_only=False)
    assert ck["global_step"] == 3 and ck["epoch"] == -1 and set(ck["data_rng"]) == {"drop", "python", "numpy", "torch"}
    train_cli.main(common + ["--save", str(b2), "--resume", str(b1)[:-3] + ".ckpt.pt"])
    wa, wb = torch.load(a, map_location="cpu"), torch.load(b2, map_location="cpu")
    assert set(wa) == set(wb)
    diff = [k for k in wa if torch.is_tensor(wa[k]) and not torch.equal(wa[k], wb[k])]
    assert not diff, diff[:5]
    moved = float((wa["decoder.conv0.weight"] - torch.load(b1, map_location="cpu")["decoder.conv0.weight"]).abs().max())
    assert moved > 0          # the resumed part really trained


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2..3]: the per-GPU shard the training bench times (16 crops of 416x544, 6x6 zones of 64 px, 34 % of
# the valid zones dropped), one training step against PyTorch autograd of the CPU oracle run in model.train() semantics.
# ---------------------------------------------------------------------------------------------------------------------
_SHARD = {}


def _shard_case():
    if not _SHARD:
        B, H, W = 16, 416, 544
        layers = spec.COMBINE1_LAYERS
        sd = weights.make_torch_state_dict(spec.model_manifest(layers))
        inp = synthetic.make_inputs(B, H, W, 6, 64, seed=5, drop_hist=0.34)                 # what bench.py --train feeds
        target = torch.from_numpy(np.stack([synthetic.make_depth(H, W, seed=50 + i, holes=0.1) for i in range(B)]))[:, None]
        offs = {"cross_atten3": (3, 5), "cross_atten2": (7, 2), "cross_atten1": (11, 20)}
        loss, pred, grads, stats = _oracle_step(layers, sd, inp, target, offs, torch.float32)   # ~1.4 GB of autograd state per crop
        _SHARD.update(layers=layers, sd=sd, inp=inp, target=target, offs=offs, loss=loss, pred=pred, grads=grads, stats=stats)
    return _SHARD


def _flat(gs, keys):
    return torch.cat([gs[k].reshape(-1).double().cpu() for k in keys])


# bounds = measured on MI355X with margin: (loss rel, pred rel-L1, full-gradient cosine, share of tensors whose rms is within 10 %)
# round 5: the 16-bit bounds sit just under what is MEASURED on the benched shard (bench line `training.fidelity_vs_f32_same_batch`,
# r5a: fp16 cosine 0.9798 / loss 4.9e-5, bf16 0.8868 / 6.4e-4) instead of far below it; how a 16-bit run OPTIMISES is the convergence test below
_SHARD_BOUNDS = {torch.float32: (2e-5, 2e-5, 0.9995, 0.99), torch.float16: (5e-4, 1e-2, 0.975, 0.8), torch.bfloat16: (3e-3, 6e-2, 0.87, 0.6)}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_config2_shard_b16_416x544_training_step_vs_oracle(dtype):
    """The benched training shard at its real size in every storage mode: loss, prediction, and the gradient of every one of the
    413 live parameter tensors (rms per tensor + cosine of the whole gradient) against float32 autograd of the oracle."""
    from cfpnet_amd.train_model import TrainNet
    c = _shard_case()
    net = TrainNet(c["sd"], c["layers"], "cuda:0", dtype=dtype)
    loss, pred, _ = net.forward_backward(c["inp"], c["target"], c["target"] > 1e-3, pos_offsets=c["offs"])
    torch.cuda.synchronize()
    g = net.grads()
    ref = c["grads"]
    live = sorted(k for k, v in ref.items() if float(v.abs().max()) > 0)
    assert len(live) >= 413 and not sorted(set(live) - set(g))[:5] and not [k for k in g if k not in ref]
    a, b = _flat(g, live), _flat(ref, live)
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    rms_ok = 0
    worst = []
    rms_max = max(float(ref[k].double().pow(2).mean().sqrt()) for k in live)
    for k in live:
        r0 = float(ref[k].double().pow(2).mean().sqrt())
        r1 = float(g[k].double().pow(2).mean().sqrt())
        # a conv bias in front of a batch-statistics BatchNorm has an exactly-zero true gradient: what autograd and the tape hold
        # there is float32 noise (rms ~1e-10 of the largest tensor's); such tensors only have to stay at noise level
        ok = abs(r1 - r0) <= 0.1 * r0 + 1e-6 * rms_max
        rms_ok += ok
        if not ok:
            worst.append((k, r0, r1))
    dl = abs(float(loss) - c["loss"]) / abs(c["loss"])
    dp = float((pred.double().cpu() - c["pred"]).abs().sum() / c["pred"].abs().sum())
    print(f"shard B=16 416x544 {dtype}: loss {float(loss):.6f} vs oracle {c['loss']:.6f} (rel {dl:.2e}); pred relL1 {dp:.2e}; "
          f"full-gradient cosine {cos:.5f}; tensors with rms within 10 %: {rms_ok}/{len(live)}; off: {worst[:4]}")
    bl, bp, bc, br = _SHARD_BOUNDS[dtype]
    assert dl < bl and dp < bp and cos > bc and rms_ok >= br * len(live)
    if dtype == torch.float32:
        for k, v in c["stats"].items():                       # running statistics after the step
            assert torch.allclose(net.buf[k].double().cpu(), v, rtol=2e-3, atol=1e-5), k
    del net
    torch.cuda.empty_cache()


def test_sixteen_bit_training_converges_like_float32_over_300_steps():
    """VERDICT r4 task 4: a gradient cosine says how one step differs, not whether a 16-bit run OPTIMISES like the float32 one the reference
    trains with (train.py:119-135, no autocast).  300 real optimisation steps (Trainer: training forward, SILog, backward, AdamW / OneCycle;
    weights.trained_like_state_dict) from the reference's own initialisation on the same six batches in the same order, positional windows
    from the same generator seed -- in float32, fp16 (the training headline) and bf16 (the type BASELINE.json names).  What is compared is
    the SILog curve averaged over the 30 steps before step 100 / 200 / 300.  Measured on MI355X (round 5): float32 1.450 / 0.414 / 0.217 from
    4.85; fp16 -2.7 % / -8.6 % / +1.0 % (i.e. BELOW float32 at step 200); bf16 +0.9 % / -1.5 % / +3.7 %.  Trajectories of a 21 M-parameter
    network diverge from a 1e-4 perturbation within a few dozen steps, so the curves agree in level, not digit for digit: the VERDICT's 2 %
    is NOT met pointwise (stated in DESIGN.md); the bound here is 15 % per checkpoint plus "every run ends below 6 % of its first loss"."""
    layers = spec.COMBINE1_LAYERS
    logs = {}
    for name, dt in (("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
        log = []
        weights.trained_like_state_dict(layers, steps=300, dtype=dt, loss_log=log)
        assert len(log) == 300 and all(np.isfinite(log))
        logs[name] = {k: float(np.mean(log[k - 30:k])) for k in (100, 200, 300)}
        logs[name]["first"] = log[0]
    print("SILog, mean of the 30 steps before step 100 / 200 / 300:", logs)
    for name in logs:
        assert logs[name][300] < 0.06 * logs[name]["first"], (name, logs)           # every mode really optimises
    for k in (100, 200, 300):
        for name in ("f16", "bf16"):
            assert abs(logs[name][k] - logs["f32"][k]) <= 0.15 * logs["f32"][k], (name, k, logs)
