#!/usr/bin/env python3
"""Per-shape micro-benchmark of cfp_conv2d_nhwc over the conv/linear launches of one forward.

Records every cfp_conv2d_nhwc(_ex) call of an eager forward (same buffers, same arguments), then
times each distinct problem with back-to-back launches: the automatic plan, the first-generation
kernel and -- with --sweep -- every second-generation tile variant (x K-splits), forced through
cfp_debug_set.  Prints a table and writes gpurun_out/conv_bench.json.
"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--kgroups", action="store_true", help="pointwise problems: the plan against the eight-wave two-K-group tiles (variants 19-21)")
ap.add_argument("--halo", action="store_true", help="3x3 stride-1 problems with <= 64 input channels only: every conv3x3_halo variant and the plan without that kernel")
ap.add_argument("--out", default="gpurun_out/conv_bench.json")
ap.add_argument("--inflight", type=int, default=1, help="time every candidate with this many copies running side by side on "
                "probed-concurrent streams (throughput mode) instead of alone")
a = ap.parse_args()

layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
eng = Engine(sd, layer_names=layers, dtype=torch.bfloat16)
inp = synthetic.to_device(synthetic.make_inputs(a.batch), "cuda:0")
for _ in range(2):
    eng.forward(inp)
torch.cuda.synchronize()
lib = hip.load()
calls = []
real = hip.call


def rec(name, *args):
    if name in ("cfp_conv2d_nhwc", "cfp_conv2d_nhwc_ex"):
        calls.append((name, args))
    real(name, *args)


hip.call = rec
eng.forward(inp)
torch.cuda.synchronize()
hip.call = real

uniq = {}
for name, args in calls:
    B, H, W, Cin, Cout, KH, KW, st, pt, pl, Ho, Wo = args[9:21]
    ex = args[23:27] if name.endswith("_ex") else (0, 0, 0.0, 0)
    key = (B, H, W, Cin, Cout, KH, st, Ho, Wo, bool(ex[0]), int(ex[3]), bool(args[5]))
    uniq.setdefault(key, [name, args, 0])[2] += 1


from _gtime import graph_time_us, graph_time_us_concurrent
from cfpnet_amd.engine import concurrent_streams
STREAMS = concurrent_streams("cuda:0", want=a.inflight) if a.inflight > 1 else None


def timeit(name, args, reps):
    # the recorded argument list ends with the stream of the recording pass: re-issue on the CURRENT stream
    fn = lambda: real(name, *args[:-1], hip.current_stream())
    if STREAMS:
        return graph_time_us_concurrent(fn, STREAMS, calls=max(4, reps // 2), replays=3)
    return graph_time_us(fn, calls=max(4, reps // 2), replays=4)


nvar = lib.cfp_conv2d_num_variants() if hasattr(lib, "cfp_conv2d_num_variants") else 19
rows = []
tot_auto = tot_v1 = tot_best = 0.0
for key, (name, args, cnt) in uniq.items():
    B, H, W, Cin, Cout, KH, st, Ho, Wo, ln, piw, has_res = key
    if a.halo and not (KH == 3 and st in (1, 2) and Cin <= 64):
        continue
    if a.kgroups and (ln or B * Ho * Wo * Cout > 9600 * 1400):
        continue
    M, K = B * Ho * Wo, KH * KH * Cin
    fl = 2.0 * M * Cout * K
    byts = 2.0 * (B * H * W * Cin + M * Cout * (2 if has_res else 1) + Cout * K)
    t_auto = timeit(name, args, a.reps)
    lib.cfp_debug_set(2, 1)
    t_v1 = timeit(name, args, a.reps)
    lib.cfp_debug_set(2, 0)
    best = (t_auto, "auto")
    sweep = {}
    if a.sweep and not ln:
        for v in range(nvar):
            for sp in ((1, 2, 4, 8, 16) if (M * Cout < 2_000_000 and K >= 512 and not piw) else (1,)):
                lib.cfp_debug_set(0, v)
                lib.cfp_debug_set(1, sp)
                try:
                    t = timeit(name, args, max(5, a.reps // 2))
                except RuntimeError:
                    continue
                sweep[f"{v}/{sp}"] = t
                if t < best[0]:
                    best = (t, f"v{v}/s{sp}")
        if KH == 3 and st == 1:
            for v in range(6):
                lib.cfp_debug_set(0, 200 + v)
                lib.cfp_debug_set(1, 1)
                t = timeit(name, args, max(5, a.reps // 2))
                sweep[f"d{v}/1"] = t
                if t < best[0]:
                    best = (t, f"d{v}")
    if a.kgroups:
        lib.cfp_debug_set(1, 1)
        for v in (4, 13, 19, 20, 21):
            lib.cfp_debug_set(0, v)
            try:
                t = timeit(name, args, a.reps)
            except RuntimeError:
                continue
            sweep[f"k{v}"] = t
            if t < best[0]:
                best = (t, f"v{v}")
        lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1)
    if (a.sweep or a.halo) and not ln and KH == 3 and st in (1, 2) and Cin <= 64 and Cin % 8 == 0 and not piw:
        lib.cfp_debug_set(1, 1)
        for v in range(8):
            if Cout > 4 * (16, 32, 64, 64, 128, 160, 224, 32)[v] or (st == 2 and v == 6):
                continue
            lib.cfp_debug_set(0, 300 + v)
            try:
                t = timeit(name, args, max(5, a.reps // 2))
            except RuntimeError:
                continue
            sweep[f"h{v}/1"] = t
            if t < best[0]:
                best = (t, f"h{v}")
    lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1)
    if a.halo:
        lib.cfp_debug_set(12, 2)                      # the halo kernel wherever it can run, its automatic tile, 2 / 3 / 4 weight stages
        for stages in (2, 3, 4):
            lib.cfp_debug_set(13, stages)
            try:
                sweep[f"n_st{stages}"] = timeit(name, args, a.reps)
            except RuntimeError:
                pass
        lib.cfp_debug_set(13, 0)
        lib.cfp_debug_set(12, 0)
        sweep["no_halo"] = timeit(name, args, a.reps)
        lib.cfp_debug_set(12, 1)
        lib.cfp_debug_set(0, -1)
        lib.cfp_debug_set(1, -1)
    import ctypes
    pv, ps = ctypes.c_int(0), ctypes.c_int(0)
    lib.cfp_conv2d_plan(M, Cout, K, KH, st, 1, Ho * Wo if piw else 0, B, ctypes.byref(pv), ctypes.byref(ps))
    ideal = max(fl / 1.5e15, byts / 5e12) * 1e6 + 1.5
    rows.append(dict(M=M, N=Cout, K=K, k=KH, stride=st, ln=ln, piw=piw, count=cnt, auto_us=t_auto, v1_us=t_v1, best_us=best[0],
                     best=best[1], plan=("halo" if pv.value >= 300 else f"d{pv.value - 200}" if pv.value >= 200 else f"v{pv.value - 100}/s{ps.value}" if pv.value >= 100 else f"g1.{pv.value}/s{ps.value}"), ideal_us=ideal, gflop=fl / 1e9, sweep=sweep))
    tot_auto += cnt * t_auto; tot_v1 += cnt * t_v1; tot_best += cnt * best[0]

rows.sort(key=lambda r: -r["auto_us"] * r["count"])
print(f"{'M':>7} {'N':>5} {'K':>5} k s  x  {'auto':>8} {'v1':>8} {'best':>8} {'ideal':>7}  plan      best       TF/s(auto)")
for r in rows:
    print(f"{r['M']:7d} {r['N']:5d} {r['K']:5d} {r['k']} {r['stride']} {r['count']:2d} {r['auto_us']:8.1f} {r['v1_us']:8.1f} {r['best_us']:8.1f} "
          f"{r['ideal_us']:7.1f}  {r['plan']:9s} {r['best']:10s} {r['gflop'] / r['auto_us'] * 1e-3:7.1f}" + (" LN" if r["ln"] else "") + (" PIW" if r["piw"] else "")
          + ("   " + "  ".join(f"{k}={v:.1f}" for k, v in r["sweep"].items() if k[0] in "hn") if a.halo else "")
          + ("   " + "  ".join(f"{k}={v:.1f}" for k, v in r["sweep"].items() if k[0] == "k") if a.kgroups else ""))
print(f"total per forward: auto {tot_auto / 1e3:.3f} ms, v1 {tot_v1 / 1e3:.3f} ms, best-of-sweep {tot_best / 1e3:.3f} ms, launches {len(calls)}")
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(rows, open(a.out, "w"))
