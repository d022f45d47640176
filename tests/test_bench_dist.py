"""The N > 1 path of bench.py on CPU: two gloo ranks, rendezvous on 127.0.0.1.  Inference does not
shard (replicas only), so what has to be right is the barrier / max-over-ranks timing and the
whole-job throughput formula."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    r, w, local, dist = bench.init_dist("gloo")
    assert (r, w, local) == (rank, world, rank) and dist is not None
    dist.barrier()
    elapsed = bench.max_over_ranks(1.0 + rank, dist, "cpu")     # rank 1 is the slow one
    dist.barrier()
    value, per_gpu = bench.job_value(w, 8, 10, elapsed)
    if rank == 0:
        torch.save({"elapsed": elapsed, "value": value, "per_gpu": per_gpu}, out)
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_throughput(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["elapsed"] == 2.0                       # MAX over ranks
    assert abs(r["value"] - 2 * 8 * 10 / 2.0) < 1e-9  # all maps of all ranks / slowest rank
    assert abs(r["per_gpu"] - r["value"] / 2) < 1e-9


def test_single_rank_needs_no_process_group(monkeypatch):
    import bench
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.init_dist("gloo") == (0, 1, 0, None)
    assert bench.max_over_ranks(3.5, None) == 3.5
    assert bench.job_value(1, 8, 10, 2.0) == (40.0, 40.0)


def test_pmc_summaries_are_read_oldest_first():
    """bench.pmc_traffic(): every committed profiles/*pmc_traffic.json is read oldest first (the un-numbered round-1 file, then r2f < ...
    < r4z), so the figure of a kernel family is the NEWEST summary that measured it.  (Round 4 found the round-1 file sorted last and
    overriding everything: the bench line reported a two-rounds-old 20.8 MB per igemm2 launch where the current kernels move 11.8 MB.)"""
    import glob
    import json
    sys.path.insert(0, ROOT)
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), key=lambda q: (os.path.basename(q)[0] == "r", os.path.basename(q)))
    assert files and os.path.basename(files[0])[0] != "r" and os.path.basename(files[-1]).startswith("r")
    fam = "igemm2<bf16,64x64,s2>"
    newest = [f for f in files if fam in json.load(open(f))][-1]
    bench._PMC_TABLE = None
    assert bench.pmc_traffic(fam) == json.load(open(newest))[fam]["hbm_bytes_per_launch"]
    assert bench.pmc_traffic_source("igemm2") == os.path.basename(newest)
    assert os.path.basename(newest) != "pmc_traffic.json"
