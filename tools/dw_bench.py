#!/usr/bin/env python3
"""Micro-benchmark of the depthwise 3x3 kernel on the encoder's shapes: automatic plan, forced
(CVB, R) configurations, and a plain row copy of the same tensor as the bandwidth yardstick."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfpnet_amd import hip, ops

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--stamps", action="store_true", help="in-kernel phase stamps of the MFMA kernel (diagnostic build path, act + 100)")
ap.add_argument("--phases", action="store_true", help="also time the kernel with its compute phase skipped (act=99)")
ap.add_argument("--ld-align", type=int, default=0, help="pad the row pitch of input and output to a multiple of this many elements (64 = whole 128-byte lines per channel block)")
ap.add_argument("--ab", action="store_true", help="also time the round-2 single-phase kernel (cfp_debug_set(6, 1)) beside the pipelined one")
ap.add_argument("--sweep-xs", action="store_true", help="sweep the sliding-window kernel's output columns per task")
ap.add_argument("--sweep-stream", action="store_true", help="sweep the pipelined kernel's rows per step S and rows per workgroup RT")
a = ap.parse_args()
lib = hip.load()
MODE = int(os.environ.get("DW_MODE", "2"))  # 2 = sliding-window kernel, 0 = pipelined (LDS-DMA) kernel, 1 = round-2 kernel
lib.cfp_debug_set(6, MODE)
DEV = "cuda:0"
B = a.batch
shapes = [(60, 80, 224, 2), (30, 40, 448, 1), (30, 40, 672, 1), (30, 40, 816, 1), (30, 40, 816, 2), (15, 20, 1392, 1)]


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _gtime import graph_time_us


def timeit(fn, reps):
    return graph_time_us(fn, calls=12, replays=4)


for (H, W, C, s) in shapes:
    Ho, Wo = -(-H // s), -(-W // s)
    pt = max((Ho - 1) * s + 3 - H, 0) // 2; pl = max((Wo - 1) * s + 3 - W, 0) // 2
    # rotate over several buffers so that back-to-back launches do not just hit in L2
    NB = 6
    ld = -(-C // a.ld_align) * a.ld_align if a.ld_align else C
    xs = [ops.Act(torch.randn(B * H * W, ld, device=DEV).to(torch.bfloat16), 0, C) for _ in range(NB)]
    outs = [ops.new_act(B * Ho * Wo, C, torch.bfloat16, DEV, ld=ld) for _ in range(NB)]
    w = torch.randn(9, C, device=DEV).to(torch.bfloat16); sc = torch.ones(C, device=DEV); sh = torch.zeros(C, device=DEV)
    part = torch.zeros(B * 64 * C + 24 * 65536, device=DEV)
    k = [0]
    def run():
        i = k[0] % NB; k[0] += 1
        ops.dwconv3x3_sum(xs[i], w, sc, sh, outs[i], part, B, H, W, s, pt, pl, Ho, Wo, hip.ACT_SILU)
    def cp():
        i = k[0] % NB; k[0] += 1
        ops.copy_rows(xs[i], outs[i], B * Ho * Wo)
    def run_nocompute():
        i = k[0] % NB; k[0] += 1
        ops.dwconv3x3_sum(xs[i], w, sc, sh, outs[i], part, B, H, W, s, pt, pl, Ho, Wo, 99)
    mb = 2.0 * (B * H * W * C + B * Ho * Wo * C) / 1e6
    t = timeit(run, a.reps); tc = timeit(cp, a.reps)
    line = f"{B}x{H}x{W}x{C} s{s}: {mb:6.1f} MB  auto {t:6.1f} us = {mb / t:5.2f} TB/s   copy_rows(out-sized) {tc:6.1f} us = {2.0 * 2 * B * Ho * Wo * C / 1e6 / tc:5.2f} TB/s"
    if a.ab:
        lib.cfp_debug_set(6, 1)
        told = timeit(run, a.reps)
        lib.cfp_debug_set(6, MODE)
        line += f"   round-2 kernel {told:6.1f} us   pipelined/copy {t / tc:4.2f}x (target <= 1.7x)   old/copy {told / tc:4.2f}x"
    if a.sweep_xs:
        for XS in (2, 3, 4, 5, 6, 7, 8, 10, 12):
            lib.cfp_debug_set(9, XS)
            try:
                tt = min(timeit(run, a.reps) for _ in range(2))
            except RuntimeError:
                continue
            line += f"\n      XS{XS}: {tt:6.1f}"
        lib.cfp_debug_set(9, 0)
    if a.sweep_stream:
        best = (t, "auto")
        for S in (1, 2, 4):
            for NW in (0, 512, 768):
                lib.cfp_debug_set(7, S); lib.cfp_debug_set(8, NW)
                try:
                    tt = timeit(run, max(5, a.reps // 3))
                except RuntimeError:
                    continue
                line += f"\n      S{S} NW{NW or 'auto'}: {tt:6.1f}"
                if tt < best[0]: best = (tt, f"S{S} NW{NW or 'auto'}")
        lib.cfp_debug_set(7, 0); lib.cfp_debug_set(8, 0)
        line += f"\n   best {best[1]} {best[0]:.1f} us"
    if a.stamps:
        ns = ops.dwconv3x3_strips(B, Ho, Wo, C, s, hip.BF16)
        part.zero_()
        torch.cuda.synchronize()
        ops.dwconv3x3_sum(xs[0], w, sc, sh, outs[0], part, B, H, W, s, pt, pl, Ho, Wo, 100 + hip.ACT_SILU)
        torch.cuda.synchronize()
        raw = part[B * ns * C: B * ns * C + 24 * 20000]
        if float(raw[5]) == 4.0:        # the sliding-window kernel: 8 floats per workgroup
            d8 = part[B * ns * C: B * ns * C + 8 * 20000].reshape(-1, 8).cpu()
            dbg = d8[d8[:, 5] == 4.0][:, :6].clone(); dbg[:, 5] = 1.0
        elif float(raw[5]) == 3.0:        # the persistent pipelined kernel: 24 floats per workgroup, per-step stamps
            d24 = raw.reshape(-1, 24).cpu()
            d24 = d24[d24[:, 5] == 3.0]
            med = lambda c: float(d24[:, c][d24[:, c] > 0].median()) if bool((d24[:, c] > 0).any()) else 0.0
            line += ("\n      per-step stamps (median cycles since workgroup start): rows landed " + " / ".join(f"{med(6 + k):.0f}" for k in range(6))
                     + "; step done " + " / ".join(f"{med(12 + k):.0f}" for k in range(6))
                     + f"; all DMA issued {med(18):.0f}, unit table built {med(19):.0f}")
            dbg = d24[:, :6].clone(); dbg[:, 5] = 1.0
        else:
            dbg = part[B * ns * C: B * ns * C + 6 * 60000].reshape(-1, 6).cpu()
        dbg = dbg[dbg[:, 5] == 1.0]
        t0, t1 = dbg[:, 3], dbg[:, 4]
        span = float((t1.max() - t0.min())) * 10.0      # 100 MHz ticks -> ns
        line += (f"\n      stamps: {dbg.shape[0]} workgroups; cycles until the first step's rows landed {dbg[:,0].median():.0f}, rest {dbg[:,1].median():.0f}, "
                 f"copy-out {dbg[:,2].median():.0f} (max {dbg[:,0].max():.0f}/{dbg[:,1].max():.0f}/{dbg[:,2].max():.0f}); "
                 f"first start -> last end {span / 1e3:.1f} us; start spread {float(t0.max() - t0.min()) * 10 / 1e3:.1f} us; "
                 f"median wg lifetime {float((t1 - t0).median()) * 10 / 1e3:.1f} us")
    if a.phases:
        line += f"   staging+copy-out only {timeit(run_nocompute, a.reps):6.1f} us"
    if a.sweep:
        best = (t, "auto")
        for cvb in (8, 16):
            for R in (1, 2, 3, 4, 5, 6, 8, 10, 15):
                if R > Ho: continue
                lib.cfp_debug_set(3, cvb); lib.cfp_debug_set(4, R)
                try:
                    tt = timeit(run, max(5, a.reps // 3))
                except RuntimeError:
                    continue
                line += f"\n      cvb{cvb} R{R}: {tt:6.1f}"
                if tt < best[0]: best = (tt, f"cvb{cvb} R{R}")
        lib.cfp_debug_set(3, 0); lib.cfp_debug_set(4, 0)
        line += f"\n   best {best[1]} {best[0]:.1f} us"
    print(line, flush=True)
