"""Data-parallel training on the HIP path: one process per rank, per-rank shards, per-replica BatchNorm statistics,
flat-gradient averaging -- the non-encoder bucket reduced on a communication stream while the RGB encoder's backward runs.
Two ranks share cuda:0 under the gloo backend (buckets staged through host memory), so the real `Trainer` step with a real
process group runs on a one-GPU box; RCCL replaces gloo on a multi-GPU node without touching this path."""
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

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(rank, B=2, H=256, W=320):
    from cfpnet_amd import synthetic
    inp = synthetic.make_inputs(B, H, W, 3, 64, seed=300 + rank, drop_hist=0.25)
    target = torch.from_numpy(np.stack([synthetic.make_depth(H, W, seed=400 + 10 * rank + i, holes=0.1) for i in range(B)]))[:, None]
    return inp, target


OFFS = {"cross_atten3": (3, 5), "cross_atten2": (7, 2), "cross_atten1": (11, 30)}


def _worker(rank, world, port, out, comm):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from cfpnet_amd import spec, synthetic, weights
    from cfpnet_amd.trainer import Trainer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    inp, target = _shard(rank)
    dinp, dtgt = synthetic.to_device(inp, "cuda:0"), target.cuda()
    tr = Trainer(sd, layers, lr=3e-4, total_steps=20, device="cuda:0", dist=dist, world=world, comm=comm)
    tr.capture(dinp, dtgt)
    assert (tr._graph2 is not None) == (comm == "overlap")
    tr.step(dinp, dtgt, pos_offsets=OFFS)
    torch.cuda.synchronize()
    g1 = tr.flat.grad.detach().cpu().clone()
    layout = [(sg.name, sg.start, sg.numel) for sg in tr.flat.segments]
    tr.step(dinp, dtgt, pos_offsets=OFFS)
    torch.cuda.synchronize()
    res = {"grad_step1": g1, "layout": layout, "params": {k: v.clone() for k, v in tr.state_dict().items() if torch.is_tensor(v)}}
    if rank == 0:
        # what each rank's gradient is on its own (no process group), same initial parameters
        singles = []
        for r in range(world):
            i2, t2 = _shard(r)
            t1 = Trainer(sd, layers, lr=3e-4, total_steps=20, device="cuda:0")
            t1._grads_to_flat(synthetic.to_device(i2, "cuda:0"), t2.cuda(), OFFS)
            torch.cuda.synchronize()
            singles.append(t1.flat.grad.detach().cpu().clone())
            assert [(sg.name, sg.start, sg.numel) for sg in t1.flat.segments] == layout
            del t1
        res["singles"] = singles
        res["initial"] = {k: v.clone() for k, v in sd.items() if torch.is_tensor(v)}
    torch.save(res, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("comm", ["overlap", "sequential"])
def test_two_ranks_on_one_gpu_train_in_lockstep(tmp_path, comm):
    from cfpnet_amd import spec
    out = str(tmp_path / "ddp")
    mp.spawn(_worker, args=(2, _free_port(), out, comm), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0", weights_only=False), torch.load(out + ".1", weights_only=False)
    # (1) the averaged gradient both ranks stepped with is the mean of the two single-rank gradients
    mean = (r0["singles"][0] + r0["singles"][1]) / 2
    assert torch.equal(r0["grad_step1"], r1["grad_step1"])
    assert torch.equal(r0["grad_step1"], mean), float((r0["grad_step1"] - mean).abs().max())
    assert float((r0["singles"][0] - r0["singles"][1]).abs().max()) > 0          # the shards really differ
    # (2) after two steps every parameter is bit-identical on the two ranks; BatchNorm statistics are per replica
    diff_params = [k for k in r0["params"] if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))
                   and not torch.equal(r0["params"][k], r1["params"][k])]
    assert not diff_params, diff_params[:5]
    assert any(not torch.equal(r0["params"][k], r1["params"][k]) for k in r0["params"] if k.endswith("running_mean"))
    # (3) the 48 dead tensors are neither reduced nor stepped; live ones moved
    dead = [k for k in r0["initial"] if spec.is_dead_param(k)]
    assert len(dead) == 48 and all(torch.equal(r0["params"][k], r0["initial"][k]) for k in dead)
    assert not torch.equal(r0["params"]["decoder.conv0.weight"], r0["initial"]["decoder.conv0.weight"])
    assert not torch.equal(r0["params"]["img_encoder.conv0.0.weight"], r0["initial"]["img_encoder.conv0.0.weight"])


def test_split_step_graphs_equal_the_single_graph():
    """Trainer.capture(split=True): the step as two HIP graphs cut where the RGB encoder's backward begins (the cut the
    gradient all-reduce overlaps) leaves exactly the parameters of the one-graph step."""
    from cfpnet_amd import spec, synthetic, weights
    from cfpnet_amd.trainer import Trainer
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    inp, target = _shard(0)
    dinp, dtgt = synthetic.to_device(inp, "cuda:0"), target.cuda()
    a = Trainer(sd, layers, lr=3e-4, total_steps=20, dtype=torch.bfloat16)
    b = Trainer(sd, layers, lr=3e-4, total_steps=20, dtype=torch.bfloat16, comm="off")
    a.capture(dinp, dtgt, wgrad_beside=False)
    b.capture(dinp, dtgt, split=True, wgrad_beside=False)
    assert a._graph2 is None and b._graph2 is not None
    for _ in range(3):
        la, _, _ = a.step(dinp, dtgt, pos_offsets=OFFS)
        lb, _, _ = b.step(dinp, dtgt, pos_offsets=OFFS)
    torch.cuda.synchronize()
    assert float(la) == float(lb) and torch.equal(a.flat.param, b.flat.param)
    for k in a.net.buf:
        assert torch.equal(a.net.buf[k], b.net.buf[k]), k


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_parameter_gradients_beside_the_backward_equal_the_single_graph(dtype):
    """Trainer.capture(wgrad_beside=True): the backward cut at the tape marks, each segment's weight / bias gradient kernels replayed as their
    own graph on a second stream beside the next segment.  Same kernels on the same operands: parameters, running statistics and
    the loss are bit-identical to the one-graph step, step after step (a block handed out twice between the two pools, or a
    gradient read after a later segment overwrote it, would show here)."""
    from cfpnet_amd import spec, synthetic, weights
    from cfpnet_amd.trainer import Trainer
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    inp, target = _shard(0)
    dinp, dtgt = synthetic.to_device(inp, "cuda:0"), target.cuda()
    a = Trainer(sd, layers, lr=3e-4, total_steps=20, dtype=dtype)
    b = Trainer(sd, layers, lr=3e-4, total_steps=20, dtype=dtype)
    a.capture(dinp, dtgt, wgrad_beside=False)
    b.capture(dinp, dtgt, wgrad_beside=True)
    assert a._segments is None and len(b._segments) == len(b.net.BACKWARD_MARKS) + 1
    for s in range(4):
        la, _, _ = a.step(dinp, dtgt, pos_offsets=OFFS)
        lb, _, _ = b.step(dinp, dtgt, pos_offsets=OFFS)
        torch.cuda.synchronize()
        assert float(la) == float(lb), (s, float(la), float(lb))
        assert torch.equal(a.flat.grad, b.flat.grad), s
    assert torch.equal(a.flat.param, b.flat.param)
    for k in a.net.buf:
        assert torch.equal(a.net.buf[k], b.net.buf[k]), k


def _silog_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from cfpnet_amd import synthetic, train_ops
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(11)
    B, Hp, Wp, Ht, Wt = 4, 60, 80, 120, 160
    pred = (torch.rand(B, 1, Hp, Wp, generator=g) * 5 + 0.5).cuda()
    tgt = torch.from_numpy(np.stack([synthetic.make_depth(Ht, Wt, seed=70 + i, holes=0.15) for i in range(B)]))[:, None].float().cuda()
    mask = tgt > 1e-3
    half = slice(rank * (B // world), (rank + 1) * (B // world))
    crit = train_ops.SILogLoss()
    crit.forward(pred[half].contiguous(), tgt[half].contiguous(), mask[half].contiguous(), interpolate=True)
    loss = crit.sync_moments(dist)
    grad = crit.backward(float(world))
    torch.cuda.synchronize()
    res = {"loss": float(loss), "grad": grad.cpu()}
    if rank == 0:
        full = train_ops.SILogLoss()
        res["full_loss"] = float(full.forward(pred, tgt, mask, interpolate=True))
        res["full_grad"] = full.backward(1.0).cpu()
    torch.save(res, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_global_batch_silog_over_two_ranks(tmp_path):
    """SILogLoss.sync_moments: two ranks hold half a batch each; after the all-reduce of (sum g, sum g^2, n) both report the loss of the
    WHOLE batch, and `backward(world)` gives world x the whole-batch gradient on the rank's own pixels (the gradient averaging over the
    ranks divides by world again) -- the reference's nn.DataParallel semantics (one loss on the gathered batch, train.py:119-123)."""
    world, out = 2, str(tmp_path / "silog")
    mp.spawn(_silog_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = [torch.load(f"{out}.{k}") for k in range(world)]
    full_loss, full_grad = r[0]["full_loss"], r[0]["full_grad"]
    for k in range(world):
        assert abs(r[k]["loss"] - full_loss) <= 2e-6 * abs(full_loss)
        want = world * full_grad[k * 2:(k + 1) * 2]
        assert float((r[k]["grad"] - want).abs().max()) <= 1e-5 * float(want.abs().max())


def _nccl_worker(rank, world, port, out):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from cfpnet_amd import spec, synthetic, weights
    from cfpnet_amd.trainer import Trainer
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    inp, target = _shard(0)
    dinp, dtgt = synthetic.to_device(inp, "cuda:0"), target.cuda()
    res = {"backend": dist.get_backend()}
    for name, kw in (("nccl_overlap", dict(dist=dist, world=1, comm="overlap")), ("nccl_sequential", dict(dist=dist, world=1, comm="sequential")),
                     ("no_dist", {})):
        tr = Trainer(sd, layers, lr=3e-4, total_steps=20, device="cuda:0", dtype=torch.bfloat16, **kw)
        tr.capture(dinp, dtgt)
        if name == "nccl_overlap":
            assert tr._graph2 is not None                    # the split step: bucket of the 10x group reduced beside the encoder's backward
        losses = []
        for _ in range(2):
            loss, _, _ = tr.step(dinp, dtgt, pos_offsets=OFFS)
            losses.append(float(loss))
        torch.cuda.synchronize()
        res[name] = {"param": tr.flat.param.detach().cpu().clone(), "grad": tr.flat.grad.detach().cpu().clone(), "loss": losses}
        del tr
        torch.cuda.empty_cache()
    torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_process_group_runs_the_overlapped_step(tmp_path):
    """First execution of the RCCL branch of the gradient averaging (`train_ops.allreduce_range` with `async_op=True` on the
    communication stream, `finish_allreduce`) on the one GPU this box has: a 1-rank "nccl" process group.  Averaging over one rank is
    the identity, so two captured steps with comm = "overlap" (split graphs, buckets reduced beside the RGB encoder's backward) and
    comm = "sequential" must leave bit for bit the parameters and gradients of the trainer without a process group.  The child is
    SPAWNED (never a re-exec of a process that has touched the GPU)."""
    out = str(tmp_path / "nccl1.pt")
    mp.spawn(_nccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    r = torch.load(out, weights_only=False)
    assert r["backend"] == "nccl"
    for mode in ("nccl_overlap", "nccl_sequential"):
        assert r[mode]["loss"] == r["no_dist"]["loss"], mode
        assert torch.equal(r[mode]["grad"], r["no_dist"]["grad"]), mode
        assert torch.equal(r[mode]["param"], r["no_dist"]["param"]), mode
    assert float(r["no_dist"]["grad"].abs().max()) > 0


def test_the_driver_command_for_two_ranks_runs_end_to_end(tmp_path):
    """The exact command the driver runs at N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`), here with
    two ranks on this box's one GPU over gloo (RCCL cannot put two ranks on one device), started as a FRESH child process before
    anything in it touches the GPU: inference replicas with in-flight slots, teardown, then the data-parallel training leg with its
    gradient all-reduce -- the whole sequence, parsed from the one JSON line rank 0 prints."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--train-batch", "4"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 only, ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"]["world_size"] == 2 and d["ranks_seen"]["backend"] == "gloo"
    assert abs(d["value"] - 2 * d["per_gpu"]) < 1e-6 * d["value"]      # whole job = sum over the ranks
    assert abs(d["value"] - 2 * 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-3 * d["value"]
    t = d["training"]
    assert t["config"]["n_gpus"] == 2 and t["config"]["global_batch"] == 8
    assert "allreduce_ms" in t and "overlap_frac" in t and t["ms_per_step"] > 0
    assert t["loss_first_step"] == t["loss_first_step"]          # finite, not NaN
