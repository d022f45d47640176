"""Training-step pieces: OneCycle schedule and flat parameter / gradient-bucket plan on CPU (incl. a two-rank gloo
all-reduce), SILog loss and AdamW HIP kernels on the GPU."""
import json
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cfpnet_amd import spec, train_ops  # noqa: E402


def test_onecycle_matches_torch_scheduler():
    """Same numbers as torch.optim.lr_scheduler.OneCycleLR configured as train.py:90-94."""
    lr, epochs, spe = 3e-4, 3, 17
    p = [torch.nn.Parameter(torch.zeros(1))]
    q = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.AdamW([{"params": p, "lr": lr / 10}, {"params": q, "lr": lr}], weight_decay=0.1, lr=lr)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, lr, epochs=epochs, steps_per_epoch=spe, cycle_momentum=True, base_momentum=0.85,
                                              max_momentum=0.95, div_factor=25, final_div_factor=100)
    mine = train_ops.OneCycle(lr, epochs * spe, 25, 100)
    for k in range(epochs * spe):
        want_lr = [g["lr"] for g in opt.param_groups]
        want_b1 = opt.param_groups[0]["betas"][0]
        got_lr, got_b1 = mine.at(k)
        assert want_lr[0] == pytest.approx(want_lr[1])          # the scalar max_lr overrides the lr/10 group
        assert got_lr == pytest.approx(want_lr[1], rel=1e-12, abs=1e-18), k
        assert got_b1 == pytest.approx(want_b1, rel=1e-12), k
        opt.step()
        sch.step()


def _manifest():
    return [(k, tuple(s)) for k, s, kind in spec.model_manifest(spec.COMBINE1_LAYERS) if kind not in ("bn_mean", "bn_var", "bn_count")]


def test_flat_layout_groups_and_dead_tail():
    named = _manifest()
    flat = train_ops.FlatParams(named, train_ops.lr_group_of(hist_encoder_10x=True))
    g0, g1, g2 = (flat.group_range[i] for i in (0, 1, 2))
    assert g0[0] == 0 and g0[1] == g1[0] and g1[1] == g2[0] == flat.live and g2[1] == flat.total
    n_dead = sum(s.numel for s in flat.segments if s.group == 2)
    assert n_dead == 388864                                       # 48 tensors, 9 D^2 + 4 D each
    assert sum(1 for s in flat.segments if s.group == 2) == 48
    n_enc = sum(s.numel for s in flat.segments if s.group == 0)
    assert all(s.name.startswith("img_encoder.") for s in flat.segments if s.group == 0) and n_enc > 12_000_000
    assert all(s.start % 4 == 0 for s in flat.segments)
    # without --hist_encoder_10x the ToF encoder trains at the encoder's rate (deltar.py:69-70)
    flat2 = train_ops.FlatParams(named, train_ops.lr_group_of(hist_encoder_10x=False))
    assert any(s.name.startswith("hist_encoder.") for s in flat2.segments if s.group == 0)
    # buckets tile the live range exactly, last-produced gradients first
    bk = flat.buckets(8 * 1024 * 1024)
    assert bk[0][1] == flat.live and bk[-1][0] == 0
    assert all(bk[i][0] == bk[i + 1][1] for i in range(len(bk) - 1))
    assert len(bk) == math.ceil(flat.live / (8 * 1024 * 1024))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    named = [("img_encoder.a", (5, 7)), ("decoder.x.transformer_path.merge.weight", (4, 4)), ("decoder.b", (1000,)), ("conv_out.0.bias", (3,))]
    flat = train_ops.FlatParams(named, train_ops.lr_group_of(True))
    flat.grad.fill_(float(rank + 1))
    dead = flat.group_range[2]
    flat.grad[dead[0]:dead[1]] = -7.0                  # must stay untouched
    train_ops.allreduce_gradients(flat, dist, world, bucket_elems=256)
    if rank == 0:
        torch.save({"grad": flat.grad.clone(), "live": flat.live, "dead": dead}, out)
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_ddp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert torch.all(r["grad"][: r["live"]] == 1.5)                       # mean of 1 and 2
    assert torch.all(r["grad"][r["dead"][0]: r["dead"][1]] == -7.0)       # dead tail is not reduced


class _FakeGroup:
    """all_reduce(sum) against a second rank whose three moments are given: enough of torch.distributed for sync_moments."""

    def __init__(self, other):
        self.other = other

    def get_backend(self, group=None):
        return "gloo"

    def all_reduce(self, t, group=None):
        t.add_(self.other.to(t.dtype))


@pytest.mark.parametrize("n_local", [0, 1])
def test_silog_sync_moments_survives_a_rank_without_valid_pixels(n_local):
    """A rank whose shard has no valid pixel reports mean = variance = NaN locally; its moments enter the global-batch loss as
    zeros, so the other ranks' loss stays finite and equals the loss over their pixels alone."""
    g = torch.tensor([0.3, -0.2, 0.5, 0.1, 0.9], dtype=torch.float64)
    other = torch.stack([g.sum(), (g * g).sum(), torch.tensor(float(len(g)), dtype=torch.float64)])
    crit = train_ops.SILogLoss()
    if n_local == 0:
        crit._stats = torch.tensor([float("nan"), float("nan"), 0.0, float("nan")])
        want_g = g
    else:
        crit._stats = torch.tensor([float("nan"), 0.7, 1.0, float("nan")])        # one pixel: a mean, no variance
        want_g = torch.cat([g, torch.tensor([0.7], dtype=torch.float64)])
    loss = crit.sync_moments(_FakeGroup(other))
    want = 10.0 * torch.sqrt(want_g.var(unbiased=True) + 0.15 * want_g.mean() ** 2)
    assert torch.isfinite(loss) and abs(float(loss) - float(want)) < 1e-5 * float(want)
    assert float(crit._stats[2]) == len(want_g)


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("masked,interp", [(True, True), (False, False)])
def test_silog_loss_matches_reference_golden_and_autograd(masked, interp, golden_dir):
    from oracle import cfpnet_oracle as O
    m = json.load(open(os.path.join(golden_dir, "misc.json")))["silog"]
    rng = np.random.default_rng(m["seed"])
    pred = torch.from_numpy(rng.uniform(0.3, 9.0, (2, 1, 26, 34)).astype(np.float32))
    gt = torch.from_numpy(rng.uniform(0.0, 9.0, (2, 1, 52, 68)).astype(np.float32))
    if masked:
        tgt, mask, want = gt, gt > 1.0, m["loss"]
    else:
        tgt, mask, want = gt[:, :, ::2, ::2].clamp(min=0.1).contiguous(), None, m["loss_nomask"]
    crit = train_ops.SILogLoss()
    loss = crit(pred.cuda(), tgt.cuda(), mask=None if mask is None else mask.cuda(), interpolate=interp)
    assert abs(float(loss) - want) < 2e-5 * want                 # the reference's own number (tests/golden/misc.json)
    p = pred.clone().requires_grad_(True)
    ref = O.silog_loss(p, tgt, mask=mask, interpolate=interp)
    ref.backward()
    grad = crit.backward(1.0).cpu()
    assert float((grad - p.grad).abs().max()) <= 2e-5 * float(p.grad.abs().max()) + 1e-9


@pytest.mark.gpu
def test_silog_loss_training_shape():
    """416x544 crops, batch 4: pred at half resolution, ~30 % invalid depth."""
    from oracle import cfpnet_oracle as O
    g = torch.Generator().manual_seed(3)
    pred = torch.rand(4, 1, 208, 272, generator=g) * 8 + 0.2
    gt = torch.rand(4, 1, 416, 544, generator=g) * 9
    gt[gt < 2.7] = 0.0
    mask = gt > 1e-3
    crit = train_ops.SILogLoss()
    loss = crit(pred.cuda(), gt.cuda(), mask=mask.cuda(), interpolate=True)
    p = pred.clone().requires_grad_(True)
    ref = O.silog_loss(p, gt, mask=mask, interpolate=True)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 2e-5 * float(ref)
    grad = crit.backward(0.5).cpu()
    assert float((grad - 0.5 * p.grad).abs().max()) <= 3e-5 * float(p.grad.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_overflow_guard_skips_a_step_with_a_non_finite_gradient(bad):
    """fp16 training's overflow guard: a gradient whose norm is not finite leaves parameters and both AdamW moments untouched (the
    step is skipped on the device, no host sync), the next finite step is applied; a guarded optimizer equals the unguarded one on
    finite gradients (the clip factor is then exactly 1)."""
    named = [("img_encoder.w", (37, 11)), ("decoder.w", (64, 33)), ("conv_out.0.weight", (1026,))]
    group = train_ops.lr_group_of(True)
    torch.manual_seed(1)
    init = {n: torch.randn(*s) for n, s in named}
    opts = []
    for guard in (True, False):
        flat = train_ops.FlatParams(named, group, device="cuda:0")
        flat.load(init)
        opts.append((flat, train_ops.FlatAdamW(flat, train_ops.OneCycle(3e-4, 10, 25, 100), weight_decay=0.1, overflow_guard=guard)))
    (fg, og), (fu, ou) = opts
    g1 = torch.randn(fg.total, device="cuda:0")
    for f, o in opts:
        f.grad.copy_(g1); o.step()
    torch.cuda.synchronize()
    assert torch.equal(fg.param, fu.param) and torch.equal(og.m, ou.m) and og.skipped_steps() == 0
    before = (fg.param.clone(), og.m.clone(), og.v.clone())
    fg.grad.copy_(g1); fg.grad[123] = bad
    og.step()
    torch.cuda.synchronize()
    assert torch.equal(fg.param, before[0]) and torch.equal(og.m, before[1]) and torch.equal(og.v, before[2])
    assert og.skipped_steps() == 1
    fg.grad.copy_(g1)
    og.step()
    torch.cuda.synchronize()
    assert not torch.equal(fg.param, before[0]) and torch.isfinite(fg.param).all() and og.skipped_steps() == 1


@pytest.mark.gpu
@pytest.mark.parametrize("clip", [None, 0.1])
def test_flat_adamw_matches_torch_adamw(clip):
    torch.manual_seed(0)
    named = [("img_encoder.w", (37, 11)), ("img_encoder.b", (5,)), ("decoder.w", (64, 33)), ("decoder.layers.1.transformer_path.merge.weight", (8, 8)),
             ("conv_out.0.weight", (1026,))]
    group = train_ops.lr_group_of(True)
    flat = train_ops.FlatParams(named, group, device="cuda:0")
    ref_p = {n: torch.nn.Parameter(torch.randn(*s)) for n, s in named}
    flat.load({n: p.detach() for n, p in ref_p.items()})
    live = [n for n, _ in named if group(n) != 2]
    opt = torch.optim.AdamW([{"params": [ref_p[n] for n in live if group(n) == 0], "lr": 3e-4 / 10},
                             {"params": [ref_p[n] for n in live if group(n) == 1], "lr": 3e-4}], weight_decay=0.1, lr=3e-4)
    total = 12
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, 3e-4, total_steps=total, cycle_momentum=True, base_momentum=0.85, max_momentum=0.95,
                                              div_factor=25, final_div_factor=100)
    mine = train_ops.FlatAdamW(flat, train_ops.OneCycle(3e-4, total, 25, 100), weight_decay=0.1, clip_grad_norm=clip)
    dead_before = flat.view("decoder.layers.1.transformer_path.merge.weight").clone()
    for it in range(total):
        for n in live:
            gr = torch.randn_like(ref_p[n]) * (0.05 if it % 2 else 3.0)
            ref_p[n].grad = gr.clone()
            flat.view(n, "grad").copy_(gr)
        if clip is not None:
            torch.nn.utils.clip_grad_norm_([ref_p[n] for n in live], clip)
        opt.step()
        sch.step()
        mine.step()
    torch.cuda.synchronize()
    for n in live:
        a, b = flat.view(n).cpu(), ref_p[n].detach()
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-7, n
    assert torch.equal(flat.view("decoder.layers.1.transformer_path.merge.weight"), dead_before)   # dead tensors are not stepped
