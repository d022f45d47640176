"""Timing helper for the micro-benchmarks: per-call time of `fn` measured inside a replayed HIP graph
(eager back-to-back launches are host-bound at ~8 us per kernel on this box; a graph node costs
~1.5 us, which is what the real forward pays)."""
import time
import torch


def graph_time_us(fn, calls=10, replays=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (replays * calls) * 1e6
