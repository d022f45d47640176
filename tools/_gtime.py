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


def graph_time_us_concurrent(fn, streams, calls=10, replays=5):
    """Per-call time of `fn` when len(streams) copies of it run side by side: one graph of `calls` launches per
    stream, all replayed together (the throughput-mode counterpart of graph_time_us)."""
    graphs = []
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(2):
                fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(calls):
                fn()
        graphs.append(g)
    def run():
        for g, st in zip(graphs, streams):
            with torch.cuda.stream(st):
                g.replay()
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (replays * calls * len(streams)) * 1e6
