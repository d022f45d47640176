#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) as per-kernel stats: calls, total, avg, min, max (us).

    python tools/prof_stats.py gpurun_out/xxx_results.db [--steps N] [--csv out.csv]
"""
import argparse, re, sqlite3, sys

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--steps", type=int, default=0, help="divide totals by this many forwards")
ap.add_argument("--csv", default=None)
ap.add_argument("--top", type=int, default=60)
a = ap.parse_args()
c = sqlite3.connect(a.db)
cols = [d[0] for d in c.execute("select * from kernels limit 1").description]
ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
agg = {}
for r in c.execute("select * from kernels"):
    d = (r[ei] - r[si]) / 1e3
    name = r[ni]
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?([A-Za-z0-9_:]+(?:<[^(]*>)?)", name)
    short = (m.group(1) if m else name)[:90]
    e = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
    e[0] += 1; e[1] += d; e[2] = min(e[2], d); e[3] = max(e[3], d)
tot = sum(e[1] for e in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
div = a.steps or 1
print(f"total kernel time {tot / 1e3:.3f} ms" + (f" = {tot / 1e3 / div:.3f} ms/step over {div} steps" if a.steps else ""))
print(f"{'kernel':90s} {'calls':>7s} {'total_us':>10s} {'avg_us':>8s} {'min':>7s} {'max':>8s} {'%':>6s}")
for k, (n, t, mn, mx) in rows[: a.top]:
    print(f"{k:90s} {n:7d} {t:10.1f} {t / n:8.2f} {mn:7.2f} {mx:8.2f} {100 * t / tot:6.2f}")
if a.csv:
    with open(a.csv, "w") as f:
        f.write("Name,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage\n")
        for k, (n, t, mn, mx) in rows:
            f.write(f"\"{k}\",{n},{t:.3f},{t / n:.3f},{mn:.3f},{mx:.3f},{100 * t / tot:.3f}\n")
