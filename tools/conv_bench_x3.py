#!/usr/bin/env python3
"""Sweep of the f16x3 implicit-GEMM tiles (csrc/conv_igemm_x3.hip) over the conv / linear launches of one float32-storage forward:
every distinct problem with the automatic plan and with every tile variant (x K-splits) forced through cfp_debug_set(0, 400 + v),
alone or with --inflight copies side by side.  Writes gpurun_out/conv_bench_x3.json (rows sorted by time x count).
    python tools/conv_bench_x3.py [--inflight 4] [--quick]"""
import argparse, ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, spec, synthetic, weights
from cfpnet_amd.engine import Engine, concurrent_streams
from _gtime import graph_time_us, graph_time_us_concurrent

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--inflight", type=int, default=1)
ap.add_argument("--quick", action="store_true", help="plan only, no sweep")
ap.add_argument("--halo", action="store_true", help="3x3 stride-1 problems only: the halo kernel's tiles against the implicit GEMM")
ap.add_argument("--out", default="gpurun_out/conv_bench_x3.json")
ap.add_argument("--only", default="", help="comma-separated implicit-GEMM variant ids to sweep instead of all (their times are printed per problem), e.g. 13,14,26,28,29,30")
a = ap.parse_args()

layers = spec.COMBINE1_LAYERS
sd = weights.make_torch_state_dict(spec.model_manifest(layers))
eng = Engine(sd, layer_names=layers, dtype=torch.float32, x3=True)
inp = synthetic.to_device(synthetic.make_inputs(a.batch), "cuda:0")
for _ in range(2):
    eng.forward(inp)
torch.cuda.synchronize()
lib = hip.load()
calls, real = [], hip.call


def rec(name, *args):
    if name == "cfp_conv2d_nhwc_ex":
        calls.append(args)
    real(name, *args)


hip.call = rec
eng.forward(inp)
torch.cuda.synchronize()
hip.call = real
uniq = {}
for args in calls:
    B, H, W, Cin, Cout, KH, KW, st, pt, pl, Ho, Wo = args[9:21]
    key = (B, H, W, Cin, Cout, KH, st, Ho, Wo, bool(args[23]), int(args[26]) & 1, bool(args[5]))
    uniq.setdefault(key, [args, 0])[1] += 1
STREAMS = concurrent_streams("cuda:0", want=a.inflight) if a.inflight > 1 else None


def timeit(args, reps):
    fn = lambda: real("cfp_conv2d_nhwc_ex", *args[:-1], hip.current_stream())
    if STREAMS:
        return graph_time_us_concurrent(fn, STREAMS, calls=max(4, reps // 2), replays=3)
    return graph_time_us(fn, calls=max(4, reps // 2), replays=4)


NV = 37
VLIST = [int(x) for x in a.only.split(",")] if a.only else list(range(NV))
rows, tot_auto, tot_best = [], 0.0, 0.0
for key, (args, cnt) in uniq.items():
    B, H, W, Cin, Cout, KH, st, Ho, Wo, ln, piw, has_res = key
    M, K = B * Ho * Wo, KH * KH * Cin
    if a.halo and not (KH == 3 and st == 1):
        continue
    t_auto = timeit(args, a.reps)
    best, sweep = (t_auto, "auto"), {}
    if not a.quick and not a.halo:
        for v in VLIST:
            for sp in ((1, 2, 4, 8) if (M * Cout < 2_000_000 and K >= 512 and not piw and v < 19) else (1,)):
                lib.cfp_debug_set(0, 400 + v); lib.cfp_debug_set(1, sp)
                try:
                    t = timeit(args, max(5, a.reps // 2))
                except RuntimeError:
                    continue
                sweep[f"{v}/{sp}"] = t
                if t < best[0]:
                    best = (t, f"v{v}/s{sp}")
        lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1)
    if KH == 3 and st == 1 and Cin % 8 == 0 and not piw and not ln and not a.only:      # conv3x3_halo_x3.hip: every tile + the implicit GEMM's plan without it
        lib.cfp_debug_set(1, 1)
        for v in list(range(10)) + [99] + ([20, 21, 22, 23, 24, 25, 36, 37, 38, 44, 45, 46] if Cin % 32 == 0 else []):
            lib.cfp_debug_set(0, 500 + v)
            try:
                t = timeit(args, max(5, a.reps // 2))
            except RuntimeError:
                continue
            sweep[f"h{v}"] = t
            if t < best[0]:
                best = (t, f"h{v}")
        lib.cfp_debug_set(0, -1); lib.cfp_debug_set(1, -1)
        lib.cfp_debug_set(24, 0)
        sweep["no_halo"] = timeit(args, a.reps)
        lib.cfp_debug_set(24, 1)
    pv, ps = ctypes.c_int(0), ctypes.c_int(0)
    lib.cfp_conv2d_plan(M, Cout, K, KH, st, hip.F32X3, Ho * Wo if piw else 0, B, ctypes.byref(pv), ctypes.byref(ps))
    rows.append(dict(M=M, N=Cout, K=K, k=KH, stride=st, ln=ln, piw=piw, res=has_res, count=cnt, auto_us=t_auto, best_us=best[0], best=best[1],
                     plan=("halo" if pv.value >= 500 else f"v{pv.value - 400}/s{ps.value}"), gflop=2.0 * M * Cout * K / 1e9, sweep=sweep))
    tot_auto += cnt * t_auto; tot_best += cnt * best[0]
rows.sort(key=lambda r: -r["auto_us"] * r["count"])
print(f"{'M':>7} {'N':>5} {'K':>5} k s  x  {'auto':>8} {'best':>8}  plan      best       TF/s(x1, auto)")
for r in rows:
    print(f"{r['M']:7d} {r['N']:5d} {r['K']:5d} {r['k']} {r['stride']} {r['count']:2d} {r['auto_us']:8.1f} {r['best_us']:8.1f}  {r['plan']:9s} {r['best']:10s} "
          f"{r['gflop'] / r['auto_us'] * 1e-3:7.1f}" + (" LN" if r["ln"] else "") + (" PIW" if r["piw"] else "")
          + ("   " + "  ".join(f"{k}={v:.1f}" for k, v in r["sweep"].items() if k[0] in "hn") if a.halo else "")
          + ("   " + "  ".join(f"v{k}={v:.1f}" for k, v in r["sweep"].items() if k.endswith("/1")) if a.only else ""))
print(f"total per forward: auto {tot_auto / 1e3:.3f} ms, best-of-sweep {tot_best / 1e3:.3f} ms, launches {len(calls)}")
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(rows, open(a.out, "w"))
