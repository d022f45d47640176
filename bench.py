#!/usr/bin/env python3
"""Headline benchmark: depth maps / second at 480x640, bf16, HIP kernels only.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one eval-mode `Deltar.forward` (RGB encoder, ToF histogram encoder, decoder with the
cross-zone fusion blocks, adaptive-bin head, `prob` output included -- exactly what the
reference returns, `/root/reference/src/models/deltar.py:34-67`) over one batch of `--batch`
synthetic 480x640 images + 8x8 ToF zones that is already resident in HBM (BASELINE.json
configs[1]).  Inference does not shard: with N > 1 every rank runs an independent replica on its
own batch (no data-path collective) and `value` is the sum over ranks = N*batch*K / max-rank-time.

The JSON line also carries
  roofline      the dominant kernel family by GPU time (timed with HIP events around every launch
                of an instrumented pass on the same stream), algorithmic FLOPs / bytes vs peak
  dw3x3         the encoder's depthwise 3x3 kernels against the HBM roofline (north_star target)
  cpu_baseline  the CPU oracle (a PyTorch-CPU restatement of the reference, `oracle/`) timed on
                this box's host cores on a bounded sample (B=1 forwards); also yields abs_rel
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E spec
RIDGE_FLOP_PER_BYTE = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)   # 312 FLOP/B: below it a kernel is bandwidth-bound by the roofline model


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one instead of replaying the HIP graph")
    ap.add_argument("--lanes", type=int, default=0, help="sub-batches of one batch run side by side, each as its own HIP graph on its own stream")
    ap.add_argument("--inflight", type=int, default=0,
                    help="whole batches kept in flight (one graph + buffers per slot, on streams probed to run concurrently). "
                         "With --lanes 0 --inflight 0 (default) the candidates are timed during set-up and the fastest is kept")
    ap.add_argument("--no-prob", action="store_true", help="skip the [B,256,H/2,W/2] prob output (not the reference contract)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-times", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the CPU baseline sample")
    ap.add_argument("--dtype", choices=("bf16", "f16"), default="bf16",
                    help="16-bit storage format of the headline line (BASELINE.json quotes bf16; f16 = IEEE half, same MFMA rate)")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE.json configs[4] instead of configs[1]: 640x960 input, 16x16 ToF zones of 40 px, per-GPU batch 2 "
                         "(16 over 8 GPUs), fp16 storage, positional tables sized for 640x960 (the reference cannot run this shape)")
    ap.add_argument("--no-f16", action="store_true", help="skip the extra fp16 measurement appended to the bf16 line")
    ap.add_argument("--no-families", action="store_true", help="skip the parity of the default mode on the other weight families (two CPU-oracle forwards each + 300 training steps)")
    ap.add_argument("--no-x3", action="store_true", help="skip the measurement of the default boundary mode (float32 storage, f16x3 matrix math)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step measurement appended to the line (N = 1 only)")
    ap.add_argument("--train", action="store_true",
                    help="measure the TRAINING step instead (BASELINE.json configs[2..3]: per-GPU batch --train-batch at 416x544, bf16, "
                         "data parallel over --gpus ranks with the RCCL gradient all-reduce inside the timed region)")
    ap.add_argument("--train-batch", type=int, default=16, help="per-GPU batch of the training-step measurement (configs[2..3]: 16)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend of --train (nccl = RCCL over xGMI; gloo stages the gradient buckets through host memory: "
                         "several ranks can then share one GPU, and with --force-dist a single rank exercises the whole DDP path)")
    ap.add_argument("--force-dist", action="store_true", help="create the process group even for one rank (measures allreduce_ms / overlap_frac at N = 1)")
    return ap.parse_args()


def kernel_times(engine, inputs, return_prob, reps=5, dw=True):
    """Instrumented eager passes: HIP events (on the launch stream) around every C-ABI call."""
    from cfpnet_amd import hip, ops
    recs = []
    dw_shapes = []
    dw_calls = []
    real_call = hip.call
    stem_w = engine.P["stem.w"].data_ptr()
    model_bytes = {}      # family -> bytes one pass fetches / writes beyond L2 under the eight-private-L2 model (see `traffic_model` in main)

    def timed_call(name, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_call(name, *a)
        e1.record()
        fam, flops, byts = name, 0.0, 0.0
        if name in ("cfp_conv2d_nhwc", "cfp_conv2d_nhwc_ex"):
            B, H, W, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo = a[9:21]   # noqa
            M = B * Ho * Wo
            cin_true = 3 if a[2] == stem_w else Cin
            flops = 2.0 * M * Cout * KH * KW * cin_true
            byts = 2.0 * (M * Cout + B * H * W * Cin + Cout * KH * KW * Cin)
            flags = a[26] if name.endswith("_ex") else 0
            piw = flags & 1
            cdt = hip.F32X3 if (flags & hip.CONV_X3) else a[22]
            esz = 4.0 if a[22] == hip.F32 else 2.0
            byts = esz * (M * Cout + B * H * W * Cin + (B if piw else 1) * Cout * KH * KW * Cin)      # per-image weights (SE-folded project GEMMs): B matrices are read
            v, sp = ops.conv2d_plan(M, Cout, KH * KW * Cin, cdt, Ho * Wo if piw else 0, B, KH, stride)
            fam = ops.conv2d_kernel_name(v, 1, a[22])       # split-K launches are folded into their tile family
            # what the launch must move past the L2s if each of the 8 XCDs (each with its own L2, each given a contiguous range of the tile
            # ids by xcd_remap, i.e. a range of ROWS) fetches the weight matrix itself: inputs and outputs once, weights once per XCD that
            # has work; per-image weights: an XCD fetches the matrices of the images its row range touches
            bm = ((ops.GEN2_TILES + ops.X3_ONLY_TILES)[v % 100][0] if v >= 400 else ops.GEN2_TILES[v - 100][0]) if (100 <= v < 200 or 400 <= v < 500) else 128
            tiles = -(-M // bm) * max(1, -(-Cout // 64))
            xcds = min(8, tiles)
            wbytes = (4.0 if cdt == hip.F32X3 else esz) * Cout * KH * KW * Cin
            wfetch = wbytes * ((B + xcds - 1) if piw else xcds)
            model_bytes[fam] = model_bytes.get(fam, 0.0) + esz * (M * Cout + B * H * W * Cin) + wfetch
        elif name in ("cfp_dwconv3x3_nhwc", "cfp_dwconv3x3_sum_nhwc", "cfp_dwconv3x3_se_nhwc"):
            B, H, W, C, stride, pt, pl, Ho, Wo = a[7:16] if name == "cfp_dwconv3x3_nhwc" else (a[8:17] if name == "cfp_dwconv3x3_sum_nhwc" else a[10:19])
            fam = "cfp_dwconv3x3_nhwc"
            flops = 2.0 * 9 * B * Ho * Wo * C
            byts = (4.0 if engine.dtype == torch.float32 else 2.0) * (B * H * W * C + B * Ho * Wo * C + 9 * C)
            dw_shapes.append((B * H * W, B * Ho * Wo, C))
            dw_calls.append((B, H, W, C, stride, pt, pl, Ho, Wo))
        elif name == "cfp_depth_head_fused":
            B, H, W = a[11:14]
            M = B * H * W
            fam = "depth_head_fused_kernel (conv3x3 128->128 + conv_out 128->256 + softmax + expectation)"
            flops = 2.0 * M * 128 * (9 * 128 + 256)
            byts = 2.0 * (M * 128 + (M * 256 if a[8] else 0) + 128 * 9 * 128 + 256 * 128) + 4.0 * M
        elif name == "cfp_dwconv_large_nhwc":
            B, H, W, C, k = a[7:12]
            flops = 2.0 * k * k * B * H * W * C
            byts = 2.0 * 2 * B * H * W * C
        recs.append((fam, e0, e1, flops, byts))
    hip.call = timed_call
    try:
        per_rep = []
        for _ in range(reps):
            recs.clear()
            engine.forward(inputs, return_prob=return_prob)
            torch.cuda.synchronize()
            per_rep.append([(fam, e0.elapsed_time(e1), flops, byts) for fam, e0, e1, flops, byts in recs])
    finally:
        hip.call = real_call
    # the launch sequence is the same in every pass: a launch's time is its MEDIAN over the passes (event pairs around
    # 10-us kernels pick up host jitter: single passes were seen 2x off)
    agg = {}
    n = min(len(r) for r in per_rep)
    for i in range(n):
        fam, _, flops, byts = per_rep[0][i]
        ts = sorted(r[i][1] for r in per_rep)
        d = agg.setdefault(fam, [0, 0.0, 0.0, 0.0])
        d[0] += reps; d[1] += ts[len(ts) // 2] * reps; d[2] += flops * reps; d[3] += byts * reps
    out = {k: dict(launches=v[0] // reps, ms=v[1] / reps, flops=v[2] / reps, bytes=v[3] / reps, model_bytes=model_bytes.get(k, 0.0) / reps) for k, v in agg.items()}
    if dw_shapes and dw:
        out["_dw3x3_copy"] = same_size_copy_ms(dw_shapes[:len(dw_shapes) // reps], engine.dtype, engine.device)
        out["_dw3x3_in_graph"] = dw3x3_in_graph(dw_calls[:len(dw_calls) // reps], engine.dtype, engine.device)
        out["_dw3x3_at_hbm_scale"] = dw3x3_at_hbm_scale(dw_calls[:len(dw_calls) // reps], engine.dtype, engine.device)
    return out


def _graph_us(fn, calls=24, replays=7):
    """Per-call time of `fn` back-to-back inside a replayed HIP graph (no per-launch host or event overhead), median of `replays`."""
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
    ts = []
    for _ in range(replays):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / calls * 1e6)
    return sorted(ts)[len(ts) // 2]


def dw3x3_at_hbm_scale(calls, dtype, dev, scale_b=16):
    """The same depthwise kernels where HBM IS the bound: every encoder shape at `scale_b` x the benched batch (tensors of 200-320 MB,
    three rotating buffer pairs = 1.3-2 GB touched per round, far beyond the 256 MiB Infinity Cache) against the same-bytes copy,
    back-to-back in a HIP graph.  At batch 8 the 13-20 MB tensors are cache resident and both the kernel and the copy are 4-12 us
    launches, i.e. the batch-8 ratio compares two latency-bound launches; this one is the bandwidth figure."""
    import math
    from cfpnet_amd import hip, ops
    shapes = {}
    for c in calls:
        shapes[c] = shapes.get(c, 0) + 1
    rows, t_k, t_c, byts = [], 0.0, 0.0, 0.0
    for (B, H, W, C, stride, pt, pl, Ho, Wo), n in shapes.items():
        Bs = B * scale_b
        esz = 4 if dtype == torch.float32 else 2
        if Bs * H * W * C * esz >= 2 ** 31 - 65536:
            Bs = int((2 ** 31 - 65536) // (H * W * C * esz))
        NB = 3
        xs = [ops.new_act(Bs * H * W, C, dtype, dev) for _ in range(NB)]
        for x in xs:
            x.buf.normal_()
        outs = [ops.new_act(Bs * Ho * Wo, C, dtype, dev) for _ in range(NB)]
        w = torch.randn(9, C, device=dev).to(dtype)
        sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        R = max(8, C // 24)
        wr = torch.randn(R, C, device=dev) / math.sqrt(C)
        f32 = dtype == torch.float32
        K = ops.dwconv3x3_strips(Bs, Ho, Wo, C, stride, ops.DT[dtype]) if f32 else ops.dwconv3x3_se_parts(Bs, Ho, Wo, C, stride, ops.DT[dtype])
        if K <= 0:
            continue
        hpart = torch.zeros(Bs * K * (C if f32 else R), device=dev)
        k = [0]

        def run():
            i = k[0] % NB; k[0] += 1
            if f32:      # float32 storage (the default f16x3 mode): depthwise + per-slot channel sums, as the in-flight plan launches it
                ops.dwconv3x3_sum(xs[i], w, sc, sh, outs[i], hpart, Bs, H, W, stride, pt, pl, Ho, Wo, hip.ACT_SILU)
            else:
                ops.dwconv3x3_se(xs[i], w, sc, sh, outs[i], wr, hpart, Bs, H, W, stride, pt, pl, Ho, Wo, hip.ACT_SILU)

        crow = (Bs * H * W + Bs * Ho * Wo) // 2
        cdst = [ops.new_act(crow, C, dtype, dev) for _ in range(NB)] if crow > Bs * Ho * Wo else outs
        csrc = [ops.new_act(crow, C, dtype, dev) for _ in range(NB)] if crow > Bs * H * W else xs

        def cp():
            i = k[0] % NB; k[0] += 1
            ops.copy_rows(csrc[i], cdst[i], crow)
        tk, tc = _graph_us(run, calls=6, replays=5), _graph_us(cp, calls=6, replays=5)
        nbytes = float(esz) * (Bs * H * W * C + Bs * Ho * Wo * C)
        cbytes = float(esz) * 2 * crow * C
        rows.append({"shape": f"{Bs}x{H}x{W}x{C} s{stride}", "launches": n, "MB": nbytes / 1e6, "us": tk, "GBps": nbytes / tk / 1e3, "copy_us": tc,
                     "copy_GBps": cbytes / tc / 1e3, "frac_of_measured_copy_rate": (nbytes / tk) / (cbytes / tc), "frac_of_hbm_peak": nbytes / tk / 1e3 / PEAK_HBM_GBS})
        t_k += n * tk; t_c += n * tc * nbytes / cbytes; byts += n * nbytes
        del xs, outs, cdst, csrc
        torch.cuda.empty_cache()
    if not rows:
        return None
    return {"GBps": byts / t_k / 1e3, "frac_of_hbm_peak": byts / t_k / 1e3 / PEAK_HBM_GBS, "copy_GBps": byts / t_c / 1e3, "frac_of_measured_copy_rate": t_c / t_k,
            "target": {"frac_of_measured_copy_rate": 0.6, "met": bool(t_c / t_k >= 0.6)}, "shapes": rows,
            "protocol": f"the forward's depthwise shapes at {scale_b} x the benched batch (200-320 MB tensors, 3 rotating buffer pairs: nothing is cache resident), "
                        "kernel and same-bytes cfp_copy_rows back-to-back in a HIP graph, median of 5 replays of 6 launches"}


def dw3x3_in_graph(calls, dtype, dev):
    """The north-star protocol for the depthwise kernels ("% of MEASURED HBM roofline"): every depthwise 3x3 launch of the forward
    (its shape, with the squeeze-excite partials it produces in the model) and a streaming copy of the same bytes, each timed
    back-to-back in a HIP graph over 6 rotating buffers (so that a launch does not find its input in L2 from the previous one).
    -> totals over the forward's launches and the per-shape table."""
    import math
    from cfpnet_amd import hip, ops
    shapes = {}
    for c in calls:
        shapes[c] = shapes.get(c, 0) + 1
    rows, t_k, t_c, byts = [], 0.0, 0.0, 0.0
    for (B, H, W, C, stride, pt, pl, Ho, Wo), n in shapes.items():
        NB = 6
        xs = [ops.new_act(B * H * W, C, dtype, dev) for _ in range(NB)]
        for x in xs:
            x.buf.normal_()
        outs = [ops.new_act(B * Ho * Wo, C, dtype, dev) for _ in range(NB)]
        w = torch.randn(9, C, device=dev).to(dtype)
        sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        R = max(8, C // 24)
        wr = torch.randn(R, C, device=dev) / math.sqrt(C)
        f32 = dtype == torch.float32
        esz = 4 if f32 else 2
        K = ops.dwconv3x3_strips(B, Ho, Wo, C, stride, ops.DT[dtype]) if f32 else ops.dwconv3x3_se_parts(B, Ho, Wo, C, stride, ops.DT[dtype])
        hpart = torch.zeros(B * max(K, 1) * (C if f32 else R), device=dev)
        k = [0]

        def run():
            i = k[0] % NB; k[0] += 1
            if f32:
                ops.dwconv3x3_sum(xs[i], w, sc, sh, outs[i], hpart, B, H, W, stride, pt, pl, Ho, Wo, hip.ACT_SILU)
            else:
                ops.dwconv3x3_se(xs[i], w, sc, sh, outs[i], wr, hpart, B, H, W, stride, pt, pl, Ho, Wo, hip.ACT_SILU)

        crow = (B * H * W + B * Ho * Wo) // 2                  # a copy that reads and writes the kernel's bytes: (rows_in + rows_out) / 2 rows each way
        cdst = [ops.new_act(crow, C, dtype, dev) for _ in range(NB)] if crow > B * Ho * Wo else outs
        csrc = [ops.new_act(crow, C, dtype, dev) for _ in range(NB)] if crow > B * H * W else xs

        def cp():
            i = k[0] % NB; k[0] += 1
            ops.copy_rows(csrc[i], cdst[i], crow)
        tk, tc = _graph_us(run), _graph_us(cp)
        nbytes = float(esz) * (B * H * W * C + B * Ho * Wo * C)
        cbytes = float(esz) * 2 * crow * C
        rows.append({"shape": f"{B}x{H}x{W}x{C} s{stride}", "launches": n, "us": tk, "GBps": nbytes / tk / 1e3, "copy_us": tc, "copy_GBps": cbytes / tc / 1e3,
                     "frac_of_measured_copy_rate": (nbytes / tk) / (cbytes / tc)})
        t_k += n * tk; t_c += n * tc * nbytes / cbytes; byts += n * nbytes
        del xs, outs
    torch.cuda.empty_cache()
    return {"us_per_step": t_k, "GBps": byts / t_k / 1e3, "copy_us_same_bytes": t_c, "copy_GBps": byts / t_c / 1e3,
            "frac_of_measured_copy_rate": t_c / t_k, "shapes": rows,
            "protocol": "back-to-back in a HIP graph, 6 rotating buffers, median of 7 replays of 24 launches; the copy is cfp_copy_rows reading and writing "
                        "(rows_in + rows_out) / 2 rows of the same channel count: the kernel's bytes"}


def same_size_copy_ms(shapes, dtype, dev, reps=5):
    """The "measured HBM roofline" of the depthwise launches: a plain streaming copy (cfp_copy_rows) moving the same number of
    bytes as each depthwise 3x3 launch (input rows + output rows, same channel count), timed with the same event-pair protocol.
    -> total ms for one pass over all the shapes."""
    from cfpnet_amd import ops
    total = 0.0
    for rows_in, rows_out, C in shapes:
        rows = (rows_in + rows_out) // 2
        src = ops.new_act(rows, C, dtype, dev)
        dst = ops.new_act(rows, C, dtype, dev)
        src.buf.normal_()
        ops.copy_rows(src, dst, rows)
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.copy_rows(src, dst, rows)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        total += sorted(ts)[len(ts) // 2]            # median, like the kernels it is compared with
    return total


_PMC_TABLE = None


def pmc_traffic(kernel_family: str):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 --pmc passes (profiles/*pmc_traffic.json, written by
    tools/pmc_traffic.py: FETCH_SIZE x2 on gfx950 for 16-byte streaming reads + WRITE_SIZE, separate passes).  Every committed summary
    is read in name order (r2f < r2k < ... < r3m < r4a ...), later ones override earlier ones kernel by kernel, so the value is the
    newest one that measured this kernel; `pmc_traffic_source` names the newest file.  None if no summary holds the kernel."""
    global _PMC_TABLE
    if _PMC_TABLE is None:
        import glob
        _PMC_TABLE = {}
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), key=lambda q: (os.path.basename(q)[0] == "r", os.path.basename(q))):      # un-numbered (round 1) files first, then r2f < ... < r4z
            try:
                for k, v in json.load(open(path)).items():
                    if isinstance(v, dict):
                        _PMC_TABLE[k] = dict(v, _file=os.path.basename(path))
            except Exception:
                pass
    e = _PMC_TABLE.get(kernel_family) or _PMC_TABLE.get(kernel_family.split(" ")[0])
    return None if e is None else e.get("hbm_bytes_per_launch")


def _train_pmc_table():
    """Newest profiles/*train_pmc.json (tools/profile_train.sh: FETCH_SIZE / WRITE_SIZE passes over the captured TRAINING step; kept apart
    from the inference summaries because the same kernel names run other shapes there)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*train_pmc.json")))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.basename(files[-1])
    except Exception:
        return None, None


def train_pmc_traffic(kernels):
    """HBM bytes per launch, averaged over the launches of the named kernel families in the training step's PMC passes; None without them."""
    tab, _ = _train_pmc_table()
    if not tab:
        return None
    byts = n = 0
    for k, v in tab.items():
        if isinstance(v, dict) and any(k.startswith(q) or (q in k) for q in kernels):
            byts += v["hbm_bytes_per_launch"] * v["launches_fetch_pass"]
            n += v["launches_fetch_pass"]
    return byts / n if n else None


def train_pmc_source():
    return _train_pmc_table()[1]


def train_pmc_step_bytes():
    """HBM bytes of ONE training step from the same passes: all kernels' bytes over the number of steps in the trace (= launches of the
    once-per-step SILog forward kernel; the pass also holds the scratch forward that fixes the kernel layouts, so this reads ~2 % high)."""
    tab, _ = _train_pmc_table()
    if not tab:
        return None
    steps = sum(v["launches_fetch_pass"] for k, v in tab.items() if isinstance(v, dict) and "silog_fwd_kernel" in k)
    tot = sum(v["hbm_bytes_per_launch"] * v["launches_fetch_pass"] for v in tab.values() if isinstance(v, dict))
    return tot / steps if steps else None


def pmc_traffic_source(kernel_family: str = ""):
    """The committed summary the figure of `kernel_family` (or of any of its tile instantiations) was read from."""
    pmc_traffic("")
    hits = sorted({v["_file"] for k, v in _PMC_TABLE.items() if kernel_family and k.startswith(kernel_family.split(" ")[0].replace("_kernel", ""))},
                  key=lambda q: (q[0] == "r", q))
    files = hits or sorted({v["_file"] for v in _PMC_TABLE.values()}, key=lambda q: (q[0] == "r", q))
    return files[-1] if files else None


def cpu_baseline(budget_s, layers, sd, batch_inputs=None):
    """The CPU oracle timed on this box's host cores (B=1 forwards, bounded sample).  With `batch_inputs` (the bench's own batch,
    host tensors) the oracle's prediction for EVERY map of that batch is also returned: the reference the 16-bit engines' rel-L1 /
    abs_rel are measured against (one untimed batched forward)."""
    from cfpnet_amd import synthetic
    from oracle import cfpnet_oracle as O
    inp = synthetic.make_inputs(1, 480, 640, 8, 56, seed=synthetic.SEED)
    # threads = cores this process may actually run on (cgroup/affinity), capped at 32: oneDNN
    # convolutions of this size stop scaling (and oversubscription is catastrophic) beyond that
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    out = O.forward(sd, inp, layer_names=layers)      # warm-up (also the abs_rel reference)
    first = time.perf_counter() - t0
    times = []
    while sum(times) + first < budget_s and len(times) < 20:
        t0 = time.perf_counter()
        O.forward(sd, inp, layer_names=layers)
        times.append(time.perf_counter() - t0)
    if not times:
        times = [first]
    times.sort()
    med = times[len(times) // 2]
    ref_batch = None
    if batch_inputs is not None:
        ref_batch = O.forward(sd, batch_inputs, layer_names=layers)[1]
    return inp, out, ref_batch, dict(value=1.0 / med, unit="maps/s", cores=torch.get_num_threads(), kind="port",
                                     sample=f"{len(times)} x B=1 480x640 full forward of the CPU oracle (PyTorch CPU fp32), median {med * 1e3:.0f} ms")


def reference_latency_ms(model, inputs, warmup=100, iters=500):
    """The reference's own latency protocol (evaluate_time.py:56-82) on the reference's own call: `model(input_data)` of the drop-in
    module (cfpnet_amd.Deltar in eval mode: a HIP-graph replay that reads these very tensors, no torch kernel).  `warmup` forwards, then
    `iters` forwards each bracketed by a device synchronize, sorted, the fastest one and the two slowest dropped, mean of the rest."""
    with torch.no_grad():
        for _ in range(warmup):
            model(inputs)
        diff = []
        for _ in range(iters):
            torch.cuda.synchronize()
            t = time.perf_counter()
            model(inputs)
            torch.cuda.synchronize()
            diff.append((time.perf_counter() - t) * 1e3)
    diff.sort()
    return sum(diff[1:-2]) / (iters - 3), diff[len(diff) // 2]


def boundary_model(sd, layers, dtype, dev, base):
    """The drop-in module around the same weights: what a user of the reference instantiates (make_model -> Deltar)."""
    from cfpnet_amd.deltar import Deltar
    import types
    ns = types.SimpleNamespace(attention_layer=list(layers), zone_sample_num=16, change_embedding=True, no_skip_inside=False, hist_encoder_10x=True)
    m = Deltar(n_bins=256, min_val=1e-3, max_val=10.0, norm="linear", args=ns, dtype=dtype, init="zeros", base_resolution=base,
               prob_dtype=None if dtype in (torch.bfloat16, torch.float16) else torch.float32)
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval()


def family_parity(layers, dev, base):
    """The default mode (f32x3) against the CPU oracle on the OTHER weight families (the benched batch covers the uniform one): the
    reference's initialisation with calibrated BatchNorm statistics, the same with a confident head, and the trained-like family
    (weights after 300 real optimisation steps) -- two maps each, relative L1 per image."""
    from cfpnet_amd import spec, synthetic, weights
    from cfpnet_amd.engine import Engine
    from oracle import cfpnet_oracle as O
    from oracle.calibrate import calibrate_bn
    out = {}
    inp = synthetic.make_inputs(2, 480, 640, 8, 56, seed=4242, drop_hist=0.2)
    dinp = synthetic.to_device(inp, dev)
    for fam in ("kaiming", "kaiming_peaked", "trained"):
        if fam == "trained":
            sd = weights.trained_like_state_dict(layers, steps=300, device=dev)
            sd.pop("__loss__", None)
        else:
            sd = calibrate_bn(weights.make_torch_state_dict(spec.model_manifest(layers, base_resolution=base), family=fam), layers)
        p0 = O.forward(sd, inp, layer_names=layers)[1]
        res = {}
        for name, kw in (("f32x3", dict(dtype=torch.float32, x3=True)), ("f16", dict(dtype=torch.float16)), ("bf16", dict(dtype=torch.bfloat16))):
            e = Engine(sd, layer_names=layers, device=dev, base_resolution=base, **kw)
            res[name] = err_vs(p0, e.forward(dinp, return_prob=False)[1])
            del e
        out[fam] = res
        torch.cuda.empty_cache()
    return out


PARITY_GATE = 1e-3          # BASELINE.json north_star: "within 1e-3 relative L1 on the predicted depth map"


def err_vs(ref, got):
    """Parity of `got` against the float32 CPU oracle on the same inputs; `gate_met` says whether THIS storage mode is inside the
    north-star tolerance.  (bf16 storage is not and cannot be: 8-bit mantissas of weights and activations put a floor of ~6e-3 under
    it, profiles/r2_precision_budget.md; fp16 storage and the float32 mode are inside.)"""
    import numpy as np
    ref, got = ref.double().cpu().numpy(), got.double().cpu().numpy()
    rel = float(np.abs(ref - got).sum() / np.abs(ref).sum())
    per_image = [float(np.abs(r - g).sum() / np.abs(r).sum()) for r, g in zip(ref, got)]
    return {"rel_l1": rel, "abs_rel": float(np.mean(np.abs(ref - got) / ref)), "rel_l1_worst_image": max(per_image),
            "gate": PARITY_GATE, "gate_met": bool(rel <= PARITY_GATE and max(per_image) <= PARITY_GATE)}


def init_dist(backend: str = "nccl", force: bool = False):
    """One process per GPU (torch.distributed.run sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns
    (rank, world, local_rank, dist-module-or-None).  `backend="gloo"` is used by the CPU tests."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1 and not force:
        return rank, world, local, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local, dist


def max_over_ranks(elapsed: float, dist, device="cpu") -> float:
    """The job's time is the slowest rank's time (contract: barrier both sides, MAX over ranks)."""
    if dist is None:
        return elapsed
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_value(world: int, batch: int, steps: int, elapsed: float):
    """Replicas only: every rank pushes its own `batch` maps per step through its own GPU, no data-path
    collective; whole-job throughput = all maps of all ranks / slowest rank's time."""
    maps = world * batch * steps
    return maps / elapsed, maps / elapsed / world


def training_fidelity(sd, layers, inp, target, dev):
    """bf16 / fp16 mixed-precision step against the float32 step on the SAME shard (the benched one): relative loss difference and
    cosine of the whole parameter gradient (all live tensors concatenated)."""
    from cfpnet_amd.train_model import TrainNet
    offs = {"cross_atten3": (3, 5), "cross_atten2": (7, 2), "cross_atten1": (11, 20)}
    out, ref = {}, None
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
        net = TrainNet(sd, layers, dev, dtype=dt)
        loss, _, _ = net.forward_backward(inp, target, target > 1e-3, pos_offsets=offs)
        torch.cuda.synchronize()
        g = net.grads()
        keys = sorted(g)
        flat = torch.cat([g[k].reshape(-1).double() for k in keys])
        if ref is None:
            ref = (float(loss), flat, keys)
        else:
            assert keys == ref[2]
            out[name] = {"loss_rel_diff": abs(float(loss) - ref[0]) / abs(ref[0]),
                         "gradient_cosine": float((flat * ref[1]).sum() / (flat.norm() * ref[1].norm()))}
        del net, g
        torch.cuda.empty_cache()
    return out


def training_kernel_times(tr, inp, target):
    """One INSTRUMENTED eager training step (HIP events around every C-ABI call, on the launch stream): time, algorithmic FLOPs and bytes of
    the convolution / linear families (forward, data gradient, weight gradient) -> the `training.roofline` object.  Launched one by one a
    call's time includes launch gaps the captured step does not pay, so short kernels read high here; the committed rocprofv3 summary of the
    captured step (profiles/*_train_step_*_kernel_stats.csv) is the cross-check."""
    from cfpnet_amd import hip
    recs, real_call = [], hip.call

    def timed_call(name, *a):
        fam, flops, byts = None, 0.0, 0.0
        if name in ("cfp_conv2d_nhwc", "cfp_conv2d_nhwc_ex", "cfp_conv2d_nhwc_moments"):
            o = 9 if name != "cfp_conv2d_nhwc_moments" else 6
            B, H, W, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo = a[o:o + 12]
            fam, M = "conv forward (implicit GEMM)", B * Ho * Wo
            flops, byts = 2.0 * M * Cout * KH * KW * Cin, 2.0 * (M * Cout + B * H * W * Cin + Cout * KH * KW * Cin)
        elif name == "cfp_conv2d_dgrad":
            B, H, W, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo = a[5:17]
            fam = "conv data gradient (implicit GEMM on the flipped weights)"
            flops, byts = 2.0 * B * H * W * Cin * KH * KW * Cout / (stride * stride), 2.0 * (B * Ho * Wo * Cout + B * H * W * Cin + Cout * KH * KW * Cin)
        elif name in ("cfp_conv2d_wgrad", "cfp_conv2d_wgrad_bias", "cfp_conv2d_wgrad_deferred"):
            o = 5 if name == "cfp_conv2d_wgrad" else 6
            B, H, W, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo = a[o:o + 12]
            fam = "conv weight gradient (conv_wgrad16 + slab reduction)"
            flops, byts = 2.0 * B * Ho * Wo * Cout * KH * KW * Cin, 2.0 * (B * Ho * Wo * Cout + B * H * W * Cin) + 4.0 * Cout * KH * KW * Cin
        if fam is None:
            fam = "other (BatchNorm / activation sweeps, depthwise, attention, loss, optimizer)"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_call(name, *a)
        e1.record()
        recs.append((fam, e0, e1, flops, byts))

    hip.call = timed_call
    try:
        tr.net.zero_grad()
        tr._grads_to_flat(inp, target, tr.draw_pos_offsets(*inp["rgb"].shape[-2:]))
        torch.cuda.synchronize()
    finally:
        hip.call = real_call
    agg = {}
    for fam, e0, e1, flops, byts in recs:
        d = agg.setdefault(fam, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        d["launches"] += 1; d["ms"] += e0.elapsed_time(e1); d["flops"] += flops; d["bytes"] += byts
    return agg


XGMI_LINK_GBPS = 153.0        # one xGMI link, one direction (MI355X_MICROARCH.md: 7 links per GPU, fully connected 8-GPU mesh)


def allreduce_plan(tr, inp, target, measure_window: bool):
    """What the data-parallel exchange of one training step moves and what it can hide behind (VERDICT r4 task 7): the per-bucket payloads
    of the flat float32 gradient in the order they are reduced, the overlap window (duration of the RGB-encoder backward, the second graph
    of the split step, measured here on one GPU), and two wire-time models per world size -- so that `allreduce_ms` of a 2 / 4 / 8-GPU
    run can be judged at first sight.  No collective runs here."""
    from cfpnet_amd import train_ops  # noqa: F401
    flat = tr.flat
    bucket_elems = 8 * 1024 * 1024
    groups, payload = [], 0
    for gi, what in ((1, "10x lr group (head, decoder, fusion blocks, ToF encoder): final after the first graph, reduced on the communication stream BESIDE the RGB-encoder backward"),
                     (0, "1x lr group (RGB encoder): reduced after the second graph, exposed")):
        lo, hi = flat.group_range[gi]
        groups.append({"group": what, "elements": int(hi - lo), "MB": (hi - lo) * 4 / 1e6,
                       "buckets_MB": [round((b1 - b0) * 4 / 1e6, 2) for b0, b1 in flat.buckets(bucket_elems, lo, hi)]})
        payload += (hi - lo) * 4
    out = {"dtype": "float32", "payload_MB": payload / 1e6, "groups_in_reduction_order": groups,
           "dead_tail_excluded_elements": int(flat.total - flat.live) if hasattr(flat, "total") else None}
    models = {}
    for n in (2, 4, 8):
        ring = 2.0 * (n - 1) / n * payload / (XGMI_LINK_GBPS * 1e9) * 1e3          # every rank sends 2 (N-1)/N x payload over ONE link (ring neighbours)
        mesh = 2.0 * payload / n / (XGMI_LINK_GBPS * 1e9) * 1e3                     # direct reduce-scatter + all-gather: payload / N per link and phase, N-1 links in parallel
        models[str(n)] = {"ring_ms_at_one_link": ring, "direct_rs_ag_ms_on_the_mesh": mesh}
    out["wire_time_models_ms"] = {"link_GBps": XGMI_LINK_GBPS, "by_world_size": models,
                                  "what": "lower bounds from bytes / link rate only (no latency, no protocol overhead); RCCL's default for 8 fully "
                                          "connected GPUs is a ring per channel over several links, so the measured figure should fall between the two"}
    if measure_window:
        if True:                                      # (the trainer is discarded right after this: its single-graph capture is replaced)
            tr.capture(inp, target, split=True)
            for _ in range(2):
                tr._graph.replay(); tr._graph2.replay()
            ts = []
            for _ in range(5):
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record(); tr._graph.replay(); e1.record(); tr._graph2.replay(); e2.record()
                torch.cuda.synchronize()
                ts.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
            ts.sort(key=lambda t: t[1])
            first, second = ts[len(ts) // 2]
            out["overlap_window"] = {"first_graph_ms": first, "rgb_encoder_backward_ms": second,
                                     "what": "the 10x group's all-reduce (first entry above) has `rgb_encoder_backward_ms` to hide behind; the 1x group's is exposed",
                                     "hidden_if_10x_allreduce_ms_below": second}
    return out


def training_step_rate(batch: int, dev, steps: int = 6, dist=None, world: int = 1, warmup: int = 2, fidelity: bool = False, dtype=None):
    """BASELINE.json configs[2..3] shape: 416x544 crops, 6x6 zones of 64 px, `batch` samples per GPU, 16-bit activations with
    float32 master parameters; one step = training forward + SILog + backward (+ gradient all-reduce) + AdamW/OneCycle, replayed as
    HIP graphs.  Headline storage type: fp16 (IEEE half, overflow-guarded optimizer) -- its gradient is the float32 one to a cosine
    of 0.98, bf16's only to 0.88 at the same speed (profiles/r3_train_fidelity.md); the bf16 step BASELINE.json names is timed
    beside it (`bf16` sub-object) when `fidelity` is set."""
    dtype = dtype or torch.float16
    dname = {torch.float16: "fp16", torch.bfloat16: "bf16"}[dtype]
    import numpy as np
    from cfpnet_amd import spec, synthetic, weights
    from cfpnet_amd.trainer import Trainer
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    H, W = 416, 544
    inp = synthetic.to_device(synthetic.make_inputs(batch, H, W, 6, 64, seed=5, drop_hist=0.34), dev)
    target = torch.from_numpy(np.stack([synthetic.make_depth(H, W, seed=50 + i, holes=0.1) for i in range(batch)]))[:, None].to(dev)
    tr = Trainer(sd, layers, lr=3e-4, total_steps=max(100, steps + warmup + 1), dtype=dtype, device=dev, dist=dist, world=world)
    tr.capture(inp, target)
    l0 = float(tr.step(inp, target)[0])
    for _ in range(max(warmup - 1, 0)):
        tr.step(inp, target)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _, _ = tr.step(inp, target)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = max_over_ranks(time.perf_counter() - t0, dist, dev) / steps
    extra = {}
    if dist is not None:
        # what the gradient averaging costs and how much of it the split step hides: the same steps with the buckets reduced after
        # the whole backward ("sequential") and without any communication ("off")
        def run_mode(mode):
            tr.comm = mode
            for _ in range(2):
                tr.step(inp, target)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                tr.step(inp, target)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            return max_over_ranks(time.perf_counter() - t1, dist, dev) / steps * 1e3
        t_seq, t_off = run_mode("sequential"), run_mode("off")
        tr.comm = "overlap"
        ar = max(t_seq - t_off, 0.0)
        extra = {"allreduce_ms": ar, "ms_per_step_sequential_allreduce": t_seq, "ms_per_step_no_allreduce": t_off,
                 "overlap_frac": (min(max((t_seq - dt * 1e3) / ar, 0.0), 1.0) if ar > 1e-3 else None),
                 "allreduce": "flat float32 gradient, 10x group (head/decoder/fusion/ToF encoder) reduced on a communication stream beside the "
                              "RGB-encoder backward (second HIP graph of the split step), 1x group after it; backend " + str(dist.get_backend())}
    elif fidelity:
        extra = {"fidelity_vs_f32_same_batch": training_fidelity(sd, layers, inp, target, dev)}
    if dist is None and fidelity:
        # roofline of the step's matrix-core families from one instrumented eager step of THIS trainer (after the timed region)
        kt = training_kernel_times(tr, inp, target)
        gem = {k: v for k, v in kt.items() if v["flops"] > 0}
        if gem:
            dom = max(gem, key=lambda k: gem[k]["ms"])
            d = gem[dom]
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
            ai = d["flops"] / max(d["bytes"], 1.0)
            gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            hbm = ai < RIDGE_FLOP_PER_BYTE
            all_fl = sum(v["flops"] for v in gem.values())
            extra["roofline"] = {"kernel": dom, "bound": "hbm" if hbm else "mfma", "achieved": gbs if hbm else tf, "peak": PEAK_HBM_GBS if hbm else PEAK_BF16_TFLOPS,
                                 "unit": "GB/s" if hbm else "TFLOP/s", "frac": gbs / PEAK_HBM_GBS if hbm else tf / PEAK_BF16_TFLOPS, "achieved_TFLOPs": tf,
                                 "flop_per_byte": ai, "launches_per_step": d["launches"], "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                                 "traffic": train_pmc_traffic(("conv_wgrad16_kernel", "wgrad_reduce_jobs_kernel") if "weight" in dom else
                                                              ("igemm2_kernel", "conv_igemm_kernel", "conv3x3_halo_kernel", "conv3x3_direct_kernel", "igemm2<", "conv_igemm<", "conv3x3_halo<", "conv3x3_direct<")),
                                 "traffic_source": train_pmc_source(),
                                 "families": {k: {"launches": v["launches"], "ms_eager_event_pairs": round(v["ms"], 3), "GFLOP": round(v["flops"] / 1e9, 1),
                                                  "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None} for k, v in kt.items()},
                                 "whole_step": {"GFLOP": all_fl / 1e9, "TFLOPs_at_the_timed_step": all_fl / dt / 1e12, "frac_of_mfma_peak": all_fl / dt / 1e12 / PEAK_BF16_TFLOPS,
                                                "hbm_bytes_pmc": train_pmc_step_bytes(),
                                                "hbm_GBps_at_the_timed_step": (train_pmc_step_bytes() / dt / 1e9) if train_pmc_step_bytes() else None,
                                                "frac_of_hbm_peak": (train_pmc_step_bytes() / dt / 1e9 / PEAK_HBM_GBS) if train_pmc_step_bytes() else None},
                                 "protocol": "one instrumented EAGER step after the timed region: HIP events around every C-ABI call on the launch stream; "
                                             "algorithmic FLOPs 2 M N K of the forward / data-gradient / weight-gradient GEMMs, 16-bit operand bytes"}
    if dtype == torch.float16:
        try:
            extra["allreduce_plan"] = allreduce_plan(tr, inp, target, measure_window=dist is None)
        except Exception as e:                        # a diagnostic appendix must never take the line down
            extra["allreduce_plan"] = {"error": repr(e)}
        extra["overflow_guard"] = {"skipped_steps": tr.opt.skipped_steps(),
                                   "what": "a step whose gradient norm is not finite is skipped on the device (cfp_grad_clip_factor / cfp_adamw_step)"}
    loss_last = float(loss)
    del tr
    torch.cuda.empty_cache()
    if fidelity and dist is None and dtype == torch.float16:
        b = training_step_rate(batch, dev, steps=steps, warmup=warmup, dtype=torch.bfloat16)
        extra["bf16"] = {"value": b["value"], "unit": "samples/s", "ms_per_step": b["ms_per_step"],
                         "note": "the storage type BASELINE.json names; same kernels, same speed, but its gradient has a cosine of ~0.88 to the "
                                 "float32 one on this shard (fidelity_vs_f32_same_batch) -- a speed number for a noisier optimisation problem"}
    return {"value": world * batch / dt, "unit": "samples/s", "ms_per_step": dt * 1e3, "dtype": f"{dname} activations, f32 master weights", **extra,
            "config": {"workload": f"batch={batch} 416x544 crops + 6x6-zone ToF, training forward + SILog + backward + AdamW/OneCycle",
                       "launch": "one HIP graph per step" + (", flat-gradient RCCL all-reduce (3 x 32 MB buckets) + AdamW after it" if world > 1 else ""),
                       "n_gpus": world, "global_batch": world * batch},
            "loss_first_step": l0, "loss_after_%d_steps" % (steps + 2): loss_last}


def main():
    a = parse()
    rank, world, local, dist = init_dist(a.backend, force=a.force_dist)
    ndev = max(1, torch.cuda.device_count())       # (counting devices does not initialise the GPU)
    local_dev = local % ndev                        # more ranks than GPUs (the 2-ranks-on-one-GPU gloo test of the driver command): they share a device
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    ranks_seen = {"world_size": dist.get_world_size() if dist else 1, "backend": dist.get_backend() if dist else None,
                  "devices_visible": ndev, "ranks_per_device": (world + ndev - 1) // ndev}

    if a.train:
        r = training_step_rate(a.train_batch, dev, steps=a.steps, dist=dist, world=world, warmup=a.warmup)
        if rank == 0:
            print(json.dumps({"metric": "training samples/sec @ 416x544, 16-bit activations (whole job: forward + SILog + backward + gradient all-reduce + AdamW)",
                              "value": r["value"], "unit": "samples/s", "per_gpu": r["value"] / world, "n_gpus": world, "steps": a.steps,
                              "warmup": a.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": r["dtype"], "data": "synthetic", "config": r["config"],
                              "loss_first_step": r["loss_first_step"],
                              **{k: r[k] for k in ("allreduce_ms", "overlap_frac", "ms_per_step_sequential_allreduce", "ms_per_step_no_allreduce", "allreduce", "allreduce_plan") if k in r}}))
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    from cfpnet_amd import spec, synthetic, weights
    from cfpnet_amd.engine import Engine
    layers = spec.COMBINE1_LAYERS
    base, zones, zone_px = spec.BASE_RESOLUTION, 8, 56
    if a.config5:
        base, zones, zone_px = (640, 960), 16, 40
        a.batch, a.height, a.width, a.dtype = 2, 640, 960, "f16"
        a.no_f16 = a.no_train = a.no_cpu_baseline = a.no_kernel_times = True     # those appendices describe configs[1]
    sd = weights.make_torch_state_dict(spec.model_manifest(layers, base_resolution=base))
    TDT = {"bf16": torch.bfloat16, "f16": torch.float16}
    engine = Engine(sd, layer_names=layers, dtype=TDT[a.dtype], device=dev, base_resolution=base)
    host_inputs = synthetic.make_inputs(a.batch, a.height, a.width, zones, zone_px, seed=synthetic.SEED + rank, image_hw=base)
    inputs = synthetic.to_device(host_inputs, dev)
    return_prob = not a.no_prob

    if dist:
        dist.barrier()          # RCCL creates its communicator streams here, before the lane choice is timed
    lane_ms = None
    if a.eager:
        a.lanes = a.lanes or 2
        step = lambda: engine.forward_lanes(inputs, a.lanes, return_prob=return_prob)
    else:
        if a.inflight > 1:
            engine.capture(inputs, return_prob=return_prob, inflight=a.inflight)
            a.inflight = len(engine._slots)
            a.lanes = 1
        elif a.lanes > 0:
            engine.capture(inputs, return_prob=return_prob, lanes=a.lanes)
        else:
            cands = (("lanes", 1), ("inflight", 2), ("inflight", 3), ("inflight", 4), ("inflight", 6))
            if world > ndev:          # ranks sharing a device share its memory and its hardware queues: fewer slots each
                free, _total = torch.cuda.mem_get_info(dev)
                cap = max(1, min(4 // ((world + ndev - 1) // ndev), int(free // (4 << 30))))
                cands = tuple(c for c in cands if c[0] == "lanes" or c[1] <= cap) or (("lanes", 1),)
            (kind, n), lane_ms = engine.capture_best(inputs, return_prob=return_prob, candidates=cands)
            a.lanes, a.inflight = (n, 1) if kind == "lanes" else (1, n)
        step = (lambda: engine.replay_async()) if a.inflight > 1 else (lambda: engine.replay())
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, dev)
    # the line's own noise estimate: the same K-step region five more times (each synchronised), per-step time of each
    repeats = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        repeats.append((time.perf_counter() - t1) / a.steps * 1e3)
    repeats.sort()

    def timed_path_pred(eng):
        """pred of the benched batch from the TIMED path: a replay of the captured graph(s) (kernel plan for batches in flight included), not
        an eager forward"""
        if a.eager:
            return eng.forward(inputs, return_prob=False)[1].clone()
        if a.inflight > 1:
            (_, p_, _), ev = eng.replay_async(inputs)
            ev.synchronize()
            return p_.clone()
        return eng.replay(inputs)[1].clone()

    pred_timed = timed_path_pred(engine)

    def timed_mode(dtype):
        """Same workload, same launch mode, same protocol in another storage mode -> (engine, seconds for a.steps steps)."""
        e2 = (Engine(sd, layer_names=layers, dtype=torch.float32, x3=True, device=dev, base_resolution=base) if dtype == "x3" else
              Engine(sd, layer_names=layers, dtype=dtype, device=dev, base_resolution=base))
        if a.eager:
            st = lambda: e2.forward_lanes(inputs, a.lanes, return_prob=return_prob)
        elif a.inflight > 1:
            e2.capture(inputs, return_prob=return_prob, inflight=a.inflight)
            st = lambda: e2.replay_async()
        else:
            e2.capture(inputs, return_prob=return_prob, lanes=a.lanes)
            st = lambda: e2.replay()
        for _ in range(a.warmup):
            st()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(a.steps):
            st()
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t2
        return e2, el2, timed_path_pred(e2)

    f16 = f32 = x3 = None
    if rank == 0 and world == 1 and (a.dtype == "bf16" or a.config5) and not a.no_x3:
        # THE DEFAULT MODE of the drop-in boundary: float32 storage, split-precision (f16 x 3) matrix math -- the mode inside the 1e-3 gate
        # on every weight family; same workload, launch mode and protocol as the headline line
        x3 = timed_mode("x3")
        x3_kt = None
        if not a.no_kernel_times:
            x3_kt = kernel_times(x3[0], inputs, return_prob, reps=3, dw=True)
        x3[0]._graph = None; x3[0]._slots = None
        x3 = (None,) + x3[1:] + (x3_kt,)
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and a.dtype == "bf16" and not a.no_f16 and not a.no_cpu_baseline:
        # IEEE-half storage (same MFMA rate, the 16-bit mode that meets the 1e-3 relative-L1 gate) and the float32 parity mode
        # (f32 MFMA, f32 storage: the mode the gate was defined for); timed here, before the instrumented passes and the CPU
        # baseline put load on the host
        f16 = timed_mode(torch.float16)
        f32 = timed_mode(torch.float32)
        f32[0]._graph = None; f32[0]._slots = None           # its in-flight slots hold ~9 GB: release before the rest
        torch.cuda.empty_cache()

    if rank == 0:
        value, per_gpu = job_value(world, a.batch, a.steps, elapsed)
        line = {
            "metric": f"depth maps/sec @ {a.height}x{a.width} {a.dtype} (whole job; per_gpu = value / n_gpus); abs_rel vs CPU oracle",
            "value": value, "unit": "maps/s", "per_gpu": per_gpu,
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"batch={a.batch} {a.height}x{a.width} RGB + {zones}x{zones}-zone ToF, eval forward incl. the prob output "
                                   f"[B,256,H/2,W/2] written in the storage dtype ({a.dtype}: 2 bytes/element; the f32x3 object of this line is the same workload with the "
                                   f"reference's float32 prob, 4 bytes/element)"
                                   if return_prob else f"batch={a.batch} {a.height}x{a.width} RGB + {zones}x{zones}-zone ToF, eval forward, prob output skipped",
                       "layers": "hist2image combine1 image x2 (CFPNet)", "launch": ("eager" if a.eager else "hipGraph replay") + (f", {a.lanes} concurrent batch lanes" if a.lanes > 1 else "")
                                 + (f", {a.inflight} batches in flight on concurrently scheduled HIP streams (one graph, buffer set and output set per slot)" if a.inflight > 1 else ""),
                       "parallelism": "replicas only" if world > 1 else "single GPU",
                       "lane_choice_ms_per_step": lane_ms},
            "ms_per_step_repeats": {"min": repeats[0], "median": repeats[2], "max": repeats[-1],
                                    "what": f"5 further repeats of the {a.steps}-step timed region, ms per step of each"},
        }
        if not a.no_kernel_times:
            kt = kernel_times(engine, inputs, return_prob)
            dw_copy_ms = kt.pop("_dw3x3_copy", None)
            dw_in_graph = kt.pop("_dw3x3_in_graph", None)
            dw_hbm = kt.pop("_dw3x3_at_hbm_scale", None)
            total_ms = sum(v["ms"] for v in kt.values())
            convs = {k: v for k, v in kt.items() if k.startswith(("conv_igemm", "igemm2", "conv3x3_direct", "depth_head_fused"))}
            # one hand-written kernel = one row: the tile shapes of igemm2_kernel are template instantiations of the same code
            groups = {}
            for k, v in convs.items():
                g = groups.setdefault(k.split("<")[0], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "traffic": 0.0, "traffic_known": True, "tiles": {}, "model": 0.0})
                g["ms"] += v["ms"]; g["flops"] += v["flops"]; g["bytes"] += v["bytes"]; g["launches"] += v["launches"]; g["model"] += v.get("model_bytes", 0.0)
                t = pmc_traffic(k)
                if t is None:
                    g["traffic_known"] = False
                else:
                    g["traffic"] += t * v["launches"]
                # which roofline bounds this tile class: algorithmic intensity (FLOP per byte of input + output + weights, 16-bit)
                # against the ridge of the machine (dense bf16 MFMA peak / HBM peak = 312 FLOP/B); most of the network's GEMMs have
                # K, N <= 256 and sit far on the bandwidth side of it
                ai = v["flops"] / max(v["bytes"], 1.0)
                gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                tfs = v["flops"] / (v["ms"] * 1e-3) / 1e12
                bound = "mfma" if ai >= RIDGE_FLOP_PER_BYTE else "hbm"
                g["tiles"][k] = {"launches_per_step": v["launches"], "avg_launch_us": v["ms"] * 1e3 / v["launches"],
                                 "achieved": tfs, "traffic": t, "flop_per_byte": ai, "achieved_GBps": gbs, "bound": bound,
                                 "frac_of_bound": tfs / PEAK_BF16_TFLOPS if bound == "mfma" else gbs / PEAK_HBM_GBS}
            names = {"igemm2": "igemm2_kernel (gen-2 16-bit implicit GEMM, all tile instantiations)",
                     "depth_head_fused_kernel (conv3x3 128->128 + conv_out 128->256 + softmax + expectation)": "depth_head_fused_kernel",
                     "conv_igemm": "conv_igemm_kernel (gen-1 implicit GEMM)", "conv3x3_direct": "conv3x3_direct_kernel"}
            dom = max(groups, key=lambda k: groups[k]["ms"])
            d = groups[dom]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            dom_ai = d["flops"] / max(d["bytes"], 1.0)
            dom_gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            dom_hbm = dom_ai < RIDGE_FLOP_PER_BYTE      # the family's own arithmetic intensity against the machine's ridge decides which roof it is under
            line["roofline"] = {"kernel": names.get(dom, dom), "bound": "hbm" if dom_hbm else "mfma",
                                "achieved": dom_gbs if dom_hbm else ach, "peak": PEAK_HBM_GBS if dom_hbm else PEAK_BF16_TFLOPS,
                                "unit": "GB/s" if dom_hbm else "TFLOP/s", "frac": dom_gbs / PEAK_HBM_GBS if dom_hbm else ach / PEAK_BF16_TFLOPS,
                                "achieved_TFLOPs": ach, "mfma_frac": ach / PEAK_BF16_TFLOPS,
                                "traffic": (d["traffic"] / d["launches"]) if d["traffic_known"] else None, "traffic_source": pmc_traffic_source(dom),
                                "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                                "traffic_model": {"bytes_per_launch": d["model"] / d["launches"] if d["model"] else None,
                                                  "what": "inputs and outputs once + the weight matrix once per XCD with work (8 private L2s; xcd_remap gives each a "
                                                          "contiguous range of rows, so activations are fetched by one L2 and weights by all eight); per-image weights: "
                                                          "the matrices of the images an XCD's rows touch.  Compare with `traffic` (PMC) and `algorithmic_bytes_per_launch`"},
                                "launches_per_step": d["launches"], "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                                "share_of_gpu_time": d["ms"] / total_ms, "flop_per_launch": d["flops"] / d["launches"],
                                "flop_per_byte": d["flops"] / max(d["bytes"], 1.0), "achieved_GBps": d["bytes"] / (d["ms"] * 1e-3) / 1e9,
                                "hbm_frac": d["bytes"] / (d["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                "tiles": d["tiles"]}
            big = max(convs, key=lambda k: convs[k]["flops"] / max(convs[k]["launches"], 1))
            e = convs[big]      # the tile family holding the largest single GEMM (the depth head's 3x3 conv)
            ach2 = e["flops"] / (e["ms"] * 1e-3) / 1e12
            line["roofline_largest_gemm"] = {"kernel": big, "bound": "mfma", "achieved": ach2, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                             "frac": ach2 / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(big), "launches_per_step": e["launches"],
                                             "avg_launch_us": e["ms"] * 1e3 / e["launches"]}
            if "cfp_dwconv3x3_nhwc" in kt:
                w = kt["cfp_dwconv3x3_nhwc"]
                gbs = w["bytes"] / (w["ms"] * 1e-3) / 1e9
                line["dw3x3"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                 "traffic": pmc_traffic("dw3x3_slide_kernel") or pmc_traffic("dw3x3_mfma_kernel"), "launches_per_step": w["launches"],
                                 "avg_launch_us": w["ms"] * 1e3 / w["launches"], "bytes_per_launch": w["bytes"] / w["launches"]}
                if dw_copy_ms:
                    # north_star: ">= 60 % of MEASURED HBM roofline on the depthwise-conv kernels": the measured roofline of a
                    # launch of this size is a streaming copy of the same bytes under the same timing protocol
                    cgbs = w["bytes"] / (dw_copy_ms * 1e-3) / 1e9
                    line["dw3x3"].update({"measured_copy_same_bytes": cgbs, "frac_of_measured_copy": gbs / cgbs,
                                          "copy_avg_launch_us": dw_copy_ms * 1e3 / w["launches"],
                                          "protocol": "median of 5 event-pair timings per launch for the kernel AND the copy; the pair adds several us "
                                                      "to both, which flatters the ratio -- see in_graph for the back-to-back figure"})
            if "dw3x3" in line and dw_in_graph:
                # the figures the north star asks for: in-graph bandwidth against the nominal peak and against the measured copy
                line["dw3x3"]["in_graph"] = dw_in_graph
                line["dw3x3"]["achieved_event_pairs"] = line["dw3x3"]["achieved"]
                line["dw3x3"]["achieved"] = dw_in_graph["GBps"]
                line["dw3x3"]["frac"] = dw_in_graph["GBps"] / PEAK_HBM_GBS
                line["dw3x3"]["frac_of_measured_copy_in_graph"] = dw_in_graph["frac_of_measured_copy_rate"]
                line["dw3x3"]["target"] = {"frac_of_measured_copy_rate": 0.6, "met": bool(dw_in_graph["frac_of_measured_copy_rate"] >= 0.6)}
            if "dw3x3" in line and dw_hbm:
                line["dw3x3"]["at_hbm_scale"] = dw_hbm
            line["kernel_ms_per_step"] = {k: round(v["ms"], 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["ms"])}
            line["kernel_ms_total"] = total_ms
        if world == 1 and not a.no_cpu_baseline:
            import numpy as np
            inp1, (e0, p0, pr0), ref_batch, cb = cpu_baseline(a.cpu_seconds, layers, sd, host_inputs)
            line["cpu_baseline"] = cb
            # parity figures on THE BENCHED BATCH (all `batch` maps), against the float32 CPU oracle
            line.update(err_vs(ref_batch, pred_timed))
            line["parity_input"] = (f"the benched batch itself: {a.batch} maps, every pixel of pred vs the CPU oracle (float32); pred taken from a replay of the "
                                    "captured graph(s) that were timed")
            for key, pair in (("f16", f16), ("f32", f32), ("f32x3", x3)):
                if pair is None:
                    continue
                e2, el2, q1 = pair[:3]
                line[key] = {"value": a.batch * a.steps / el2, "unit": "maps/s", "ms_per_step": el2 / a.steps * 1e3, "dtype": key,
                             "launch": "same as the headline line", **err_vs(ref_batch, q1)}
            if x3 is not None:
                line["f32x3"]["note"] = ("THE DEFAULT of the drop-in boundary (Deltar / make_model): float32 storage, every convolution / linear layer as "
                                         "A_hi W_hi + A_hi W_lo + A_lo W_hi on v_mfma_f32_16x16x32_f16 (csrc/conv_igemm_x3.hip), float32 prob from the kernels; "
                                         "inside the 1e-3 gate on every weight family (families below; tests/test_forward_gpu.py gates every image)")
                fams = {"uniform": {"f32x3": {k: line["f32x3"][k] for k in ("rel_l1", "rel_l1_worst_image", "gate_met")},
                                    "bf16": {k: line[k] for k in ("rel_l1", "rel_l1_worst_image", "gate_met")}}}
                if f16 is not None:
                    fams["uniform"]["f16"] = {k: line["f16"][k] for k in ("rel_l1", "rel_l1_worst_image", "gate_met")}
                if not a.no_families:
                    fams.update(family_parity(layers, dev, base))
                line["f32x3"]["families"] = fams
                line["f32x3"]["gate_met_on_every_family"] = all(v["f32x3"]["gate_met"] for v in fams.values())
                if x3[3] is not None:
                    kt3 = x3[3]
                    dw3 = {k: kt3.pop(k, None) for k in ("_dw3x3_copy", "_dw3x3_in_graph", "_dw3x3_at_hbm_scale")}
                    if "cfp_dwconv3x3_nhwc" in kt3 and dw3["_dw3x3_in_graph"]:
                        # the depthwise 3x3 launches of the DEFAULT mode (float32 storage: dw3x3_rows_kernel, csrc/dw3x3_rows.hip) under the same
                        # two protocols as the 16-bit object at the top level: back-to-back in a graph at the benched batch, and at HBM scale
                        w3 = kt3["cfp_dwconv3x3_nhwc"]
                        ig, hb = dw3["_dw3x3_in_graph"], dw3["_dw3x3_at_hbm_scale"]
                        line["f32x3"]["dw3x3"] = {
                            "kernel": "dw3x3_rows_kernel (float32 storage, register-sliding rows, no LDS)", "bound": "hbm", "achieved": ig["GBps"],
                            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ig["GBps"] / PEAK_HBM_GBS, "traffic": pmc_traffic("dw3x3_rows_kernel"),
                            "launches_per_step": w3["launches"], "bytes_per_launch": w3["bytes"] / w3["launches"],
                            "avg_launch_us_event_pairs": w3["ms"] * 1e3 / w3["launches"], "in_graph": ig, "at_hbm_scale": hb,
                            "frac_of_measured_copy_in_graph": ig["frac_of_measured_copy_rate"],
                            "target": {"frac_of_measured_copy_rate": 0.6, "met_in_graph_at_benched_batch": bool(ig["frac_of_measured_copy_rate"] >= 0.6),
                                       "met": bool(hb and hb["frac_of_measured_copy_rate"] >= 0.6)}}
                    g3 = {k: v for k, v in kt3.items() if k.startswith("igemm_x3")}
                    if g3:
                        ms3 = sum(v["ms"] for v in g3.values()); fl3 = sum(v["flops"] for v in g3.values()); by3 = sum(v["bytes"] for v in g3.values())
                        n3 = sum(v["launches"] for v in g3.values())
                        big3 = max(g3, key=lambda k: g3[k]["flops"] / max(g3[k]["launches"], 1))
                        # which roof: the family's algorithmic intensity (2 M N K over float32 inputs + outputs + weights) against the ridge,
                        # the rule of the headline's roofline object above
                        x3_hbm = fl3 / max(by3, 1.0) < RIDGE_FLOP_PER_BYTE
                        x3_tfs, x3_gbs = fl3 / (ms3 * 1e-3) / 1e12, by3 / (ms3 * 1e-3) / 1e9
                        line["f32x3"]["roofline"] = {
                            "kernel": "igemm_x3_kernel (float32 storage, 3 half MFMAs per product block; all tile instantiations)",
                            "bound": "hbm" if x3_hbm else "mfma", "achieved": x3_gbs if x3_hbm else x3_tfs, "peak": PEAK_HBM_GBS if x3_hbm else PEAK_BF16_TFLOPS,
                            "unit": "GB/s" if x3_hbm else "TFLOP/s", "frac": x3_gbs / PEAK_HBM_GBS if x3_hbm else x3_tfs / PEAK_BF16_TFLOPS,
                            "achieved_TFLOPs": x3_tfs, "mfma_frac": x3_tfs / PEAK_BF16_TFLOPS, "algorithmic_bytes_per_launch": by3 / n3,
                            "traffic_source": pmc_traffic_source("igemm_x3_kernel"),
                            "mfma_issue_TFLOPs": 3 * fl3 / (ms3 * 1e-3) / 1e12, "mfma_issue_frac": 3 * fl3 / (ms3 * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                            "what": "achieved = ALGORITHMIC FLOPs (2 M N K) per second; the kernel issues three matrix instructions per product block, "
                                    "so its own ceiling is a third of the 16-bit peak (mfma_issue_* counts all three)",
                            "launches_per_step": n3, "avg_launch_us": ms3 * 1e3 / n3, "share_of_gpu_time": ms3 / sum(v["ms"] for v in kt3.values()),
                            "flop_per_byte": fl3 / max(by3, 1.0), "achieved_GBps": by3 / (ms3 * 1e-3) / 1e9, "traffic": pmc_traffic("igemm_x3_kernel"),
                            "largest_gemm": {"kernel": big3, "achieved": g3[big3]["flops"] / (g3[big3]["ms"] * 1e-3) / 1e12,
                                             "avg_launch_us": g3[big3]["ms"] * 1e3 / g3[big3]["launches"], "launches_per_step": g3[big3]["launches"]}}
                    line["f32x3"]["kernel_ms_per_step"] = {k: round(v["ms"], 4) for k, v in sorted(kt3.items(), key=lambda kv: -kv[1]["ms"])}
            # what the driver's parsed line keeps is `config`: say there which number is inside the tolerance
            line["config"]["parity"] = {k: line[k] for k in ("rel_l1", "abs_rel", "rel_l1_worst_image", "gate", "gate_met") if k in line}
            line["config"]["parity"]["what"] = f"{a.dtype} storage (the headline `value`) vs the float32 CPU oracle on the benched batch; gate = north-star 1e-3 relative L1"
            if x3 is not None:
                line["config"]["compliant_default"] = {
                    "dtype": "f32x3", "value": line["f32x3"]["value"], "unit": "maps/s", "ms_per_step": line["f32x3"]["ms_per_step"],
                    "rel_l1": line["f32x3"]["rel_l1"], "abs_rel": line["f32x3"]["abs_rel"], "gate_met": line["f32x3"]["gate_met"],
                    "gate_met_on_every_family": line["f32x3"]["gate_met_on_every_family"],
                    "what": "the drop-in boundary's DEFAULT mode (Deltar / make_model): float32 storage, f16 x 3 split-precision matrix math, float32 prob"}
                if not line.get("gate_met", False):
                    line["config"]["workload"] += (f"; {a.dtype} storage = OPT-IN SPEED MODE OUTSIDE the 1e-3 gate (rel_l1 {line['rel_l1']:.2e}); "
                                                   f"gate-compliant default = f32x3 {line['f32x3']['value']:.0f} maps/s (config.compliant_default)")
            if f32 is not None:
                line["f32"]["note"] = ("float32 parity mode: f32 storage, v_mfma_f32_16x16x4_f32 -- the mode that meets the north-star 1e-3 relative-L1 gate "
                                       "by three orders of magnitude, timed under the same protocol")
            # the reference's own latency protocol (evaluate_time.py:56-82: 100 warm-up, 500 timed, trimmed mean), one graph per forward
            lat = {}
            engine._graph = None; engine._slots = None          # the throughput slots are done: their memory goes back before the latency modules capture
            torch.cuda.empty_cache()
            for mode, mdt in ((a.dtype, TDT[a.dtype]), ("f32x3", "f32x3")):
                if mode == "f32x3" and x3 is None:
                    continue
                model = boundary_model(sd, layers, mdt, dev, base)
                for bsz in (1, a.batch):
                    li = synthetic.to_device(synthetic.make_inputs(bsz, a.height, a.width, zones, zone_px, seed=synthetic.SEED, image_hw=base), dev)
                    model.eval_static_outputs = True       # the latency loop's opt-in (evaluate_time.py of this repo sets it too): results are the ring's buffers
                    mean_ms, med_ms = reference_latency_ms(model, li)
                    tgt = lat if mode == a.dtype else lat.setdefault("f32x3", {})
                    tgt[f"latency_b{bsz}"] = {"ms": mean_ms, "median_ms": med_ms, "maps_per_s": bsz / mean_ms * 1e3}
                    model.eval_static_outputs = False      # the module's DEFAULT: fresh (edges, pred, prob) tensors per call like the reference (three clone kernels)
                    mean_ms, med_ms = reference_latency_ms(model, li, warmup=20, iters=100)
                    tgt[f"latency_b{bsz}"]["fresh_outputs_ms"] = mean_ms
                del model
                torch.cuda.empty_cache()
            lat["protocol"] = ("evaluate_time.py:56-82 on the reference's own call: 100 warm-up + 500 timed `model(input_data)` of the drop-in module "
                               "(cfpnet_amd.Deltar, eval mode, eval_static_outputs=True: one HIP-graph replay per forward reading the caller's tensors, results in the module's "
                               "output ring; fresh_outputs_ms = the module's default, which clones the three results per call), each synchronised, min 1 / max 2 "
                               f"dropped, mean; top level = {a.dtype} storage with prob as stored, f32x3 = the module's default mode (float32 prob)")
            line["latency"] = lat
        if world == 1 and not a.no_train and not a.no_cpu_baseline:
            line["training"] = training_step_rate(a.train_batch, dev, fidelity=True)
    train_multi = None
    if world > 1 and not a.no_train and not a.config5:
        # the data-parallel TRAINING step on all ranks (BASELINE.json configs[2..3]: per-GPU batch 16, RCCL all-reduce of the flat
        # gradient inside the timed region): inference is replicas only, so this object is where the 1 -> N scaling of the north star
        # ("linear 1->8 GPU DDP scaling >= 0.9x") is measured.  Every rank takes part; rank 0 reports.
        engine._graph = None; engine._slots = None; engine._plans.clear()
        engine = None
        torch.cuda.empty_cache()
        train_multi = training_step_rate(a.train_batch, dev, dist=dist, world=world)
        train_multi["per_gpu"] = train_multi["value"] / world
    if rank == 0:
        if train_multi is not None:
            line["training"] = train_multi
        if x3 is not None and "f32x3" not in line:      # --config5: the default mode timed under the same protocol (parity: tests/test_forward_gpu.py's config5 case)
            line["f32x3"] = {"value": a.batch * a.steps / x3[1], "unit": "maps/s", "ms_per_step": x3[1] / a.steps * 1e3, "dtype": "f32x3",
                             "launch": "same as the headline line", "note": "the drop-in boundary's default numerics (float32 storage, f16x3 matrix math, float32 prob)"}
        print(json.dumps(line))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
