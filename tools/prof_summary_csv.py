#!/usr/bin/env python3
"""Per-kernel stats from a rocprofv3 `--kernel-trace --output-format csv` run (the raw *_kernel_trace.csv is tens of MB; this is
what gets committed under profiles/).

    python tools/prof_summary_csv.py gpurun_out/prof_r2/r2_kernel_trace.csv --steps 60 --csv profiles/r2_bench_b8_kernel_stats.csv
"""
import argparse, collections, csv, re, sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--steps", type=int, default=0, help="forwards in the trace (warm-up + timed + set-up passes): per-step figures are totals / steps")
ap.add_argument("--csv", default=None)
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--exclude", default="spin_kernel", help="regex of set-up kernels kept out of the totals and shares: `at::cuda::spin_kernel` is "
                "torch.cuda._sleep, launched by engine.concurrent_streams() while it probes which HIP streams run side by side (before any timed region)")
a = ap.parse_args()
csv.field_size_limit(1 << 30)
agg = collections.OrderedDict()
skipped = {}
with open(a.trace) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = name.replace("(anonymous namespace)::", "")          # at::cuda::(anonymous namespace)::spin_kernel -> at::cuda::spin_kernel
        m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]*>)?)", name)
        short = (m.group(1) if m else name)[:110]
        if a.exclude and re.search(a.exclude, short):
            skipped[short] = skipped.get(short, 0.0) + d
            continue
        e = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
        e[0] += 1; e[1] += d; e[2] = min(e[2], d); e[3] = max(e[3], d)
tot = sum(e[1] for e in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f"total kernel time {tot / 1e3:.3f} ms, {sum(e[0] for e in agg.values())} launches" + (f" = {tot / 1e3 / a.steps:.3f} ms and {sum(e[0] for e in agg.values()) / a.steps:.1f} launches per forward over {a.steps} forwards" if a.steps else ""))
for k, t in skipped.items():
    print(f"excluded from totals and shares (set-up, outside every timed region): {k} {t / 1e3:.3f} ms")
print(f"{'kernel':110s} {'calls':>7s} {'total_us':>11s} {'avg_us':>8s} {'min':>7s} {'max':>8s} {'%':>6s}")
for k, (n, t, mn, mx) in rows[: a.top]:
    print(f"{k:110s} {n:7d} {t:11.1f} {t / n:8.2f} {mn:7.2f} {mx:8.2f} {100 * t / tot:6.2f}")
if a.csv:
    with open(a.csv, "w") as f:
        f.write("Name,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage\n")
        for k, (n, t, mn, mx) in rows:
            f.write(f"\"{k}\",{n},{t:.3f},{t / n:.3f},{mn:.3f},{mx:.3f},{100 * t / tot:.3f}\n")
