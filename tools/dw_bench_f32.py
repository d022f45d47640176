#!/usr/bin/env python3
"""Depthwise 3x3 in FLOAT32 storage (the default f16x3 mode's kernel) on the encoder's six shapes: the register-sliding kernel
(dw3x3_rows.hip, round 5) against the round-1 LDS-strip kernel and against a same-bytes copy, at batch 8 and at HBM scale (16 x the
batch, rotating buffers), optionally sweeping the run length R.

    python tools/dw_bench_f32.py [--batch 8] [--scale 16] [--sweep-r]
"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--scale", type=int, default=16, help="second pass at this multiple of the batch (0 = skip)")
ap.add_argument("--sweep-r", action="store_true")
ap.add_argument("--json", default=None)
a = ap.parse_args()
lib = hip.load()
DEV = "cuda:0"
shapes = [(60, 80, 224, 2), (30, 40, 448, 1), (30, 40, 672, 1), (30, 40, 816, 1), (30, 40, 816, 2), (15, 20, 1392, 1)]
COUNT = {(60, 80, 224, 2): 1, (30, 40, 448, 1): 4, (30, 40, 672, 1): 6, (30, 40, 816, 1): 1, (30, 40, 816, 2): 1, (15, 20, 1392, 1): 11}   # launches per forward
out = {}
for B in [a.batch] + ([a.batch * a.scale] if a.scale else []):
    tot = {"new": 0.0, "old": 0.0, "copy": 0.0, "bytes": 0.0}
    for (H, W, C, s) in shapes:
        Ho, Wo = -(-H // s), -(-W // s)
        pt = max((Ho - 1) * s + 3 - H, 0) // 2; pl = max((Wo - 1) * s + 3 - W, 0) // 2
        nbytes = 4.0 * (B * H * W * C + B * Ho * Wo * C)
        NB = max(2, min(6, int(3e9 // nbytes)))            # rotate: back-to-back launches must not hit in L2 / MALL
        xs = [ops.Act(torch.randn(B * H * W, C, device=DEV), 0, C) for _ in range(NB)]
        outs = [ops.new_act(B * Ho * Wo, C, torch.float32, DEV) for _ in range(NB)]
        w = torch.randn(9, C, device=DEV); sc = torch.rand(C, device=DEV) + 0.5; sh = torch.randn(C, device=DEV)
        part = torch.zeros(B * 4096 * C // 8 + 1024, device=DEV)
        k = [0]
        def run():
            i = k[0] % NB; k[0] += 1
            ns = ops.dwconv3x3_strips(B, Ho, Wo, C, s, ops.DT[torch.float32])
            assert B * ns * C <= part.numel(), (ns, part.numel())
            ops.dwconv3x3_sum(xs[i], w, sc, sh, outs[i], part, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
        # same-bytes copy: reads the input, writes the output extent (one fused copy launch moving in + out bytes)
        src = [torch.empty(int(nbytes // 8), device=DEV) for _ in range(NB)]
        dst = [torch.empty(int(nbytes // 8), device=DEV) for _ in range(NB)]
        def cp():
            i = k[0] % NB; k[0] += 1
            dst[i].copy_(src[i])
        calls = 12 if B <= 16 else 6
        lib.cfp_debug_set(10, 1); lib.cfp_debug_set(11, 0)
        t_new = graph_time_us(run, calls=calls, replays=4)
        lib.cfp_debug_set(10, 0)
        t_old = graph_time_us(run, calls=calls, replays=4)
        lib.cfp_debug_set(10, 1)
        t_cp = graph_time_us(cp, calls=calls, replays=4)
        line = (f"B={B} {H}x{W}x{C} s{s}: {nbytes / 1e6:7.1f} MB  rows {t_new:7.1f} us = {nbytes / t_new / 1e6:5.2f} TB/s   lds-strip {t_old:7.1f} us = "
                f"{nbytes / t_old / 1e6:5.2f} TB/s   copy {t_cp:7.1f} us = {nbytes / t_cp / 1e6:5.2f} TB/s   rows/copy {t_cp / t_new:4.2f}")
        if a.sweep_r:
            sw = []
            for R in (1, 2, 3, 4, 5, 6, 8, 10, 15, 20, 30):
                if R > Ho:
                    continue
                lib.cfp_debug_set(11, R)
                sw.append(f"R{R}:{graph_time_us(run, calls=calls, replays=3):.1f}")
            lib.cfp_debug_set(11, 0)
            line += "   " + " ".join(sw)
        print(line, flush=True)
        n = COUNT[(H, W, C, s)]
        tot["new"] += n * t_new; tot["old"] += n * t_old; tot["copy"] += n * t_cp; tot["bytes"] += n * nbytes
        out[f"B{B}_{H}x{W}x{C}_s{s}"] = {"MB": nbytes / 1e6, "rows_us": t_new, "lds_strip_us": t_old, "copy_us": t_cp, "frac_of_copy": t_cp / t_new}
        del xs, outs, src, dst
        torch.cuda.empty_cache()
    print(f"B={B} forward-weighted (24 launches): rows {tot['new']:.0f} us = {tot['bytes'] / tot['new'] / 1e6:.2f} TB/s, lds-strip {tot['old']:.0f} us, "
          f"copy {tot['copy']:.0f} us = {tot['bytes'] / tot['copy'] / 1e6:.2f} TB/s -> {tot['copy'] / tot['new']:.2f} of the copy", flush=True)
    out[f"B{B}_forward"] = {"rows_us": tot["new"], "lds_strip_us": tot["old"], "copy_us": tot["copy"], "TBps": tot["bytes"] / tot["new"] / 1e6,
                            "copy_TBps": tot["bytes"] / tot["copy"] / 1e6, "frac_of_copy": tot["copy"] / tot["new"]}
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
