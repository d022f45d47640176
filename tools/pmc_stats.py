#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one or more files)."""
import csv, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?([A-Za-z0-9_:]+(?:<[^(]*>)?)", name)
            short = (m.group(1) if m else name)[:70]
            grid = r.get("Grid_Size", "")
            e = agg[(short, grid)][r["Counter_Name"]]
            e[0] += 1; e[1] += float(r["Counter_Value"])
for (k, grid), cs in sorted(agg.items()):
    if "elementwise" in k or "Fill" in k: continue
    print(f"{k}  grid={grid}")
    for c, (n, v) in sorted(cs.items()):
        print(f"    {c:24s} calls {n:5d}  avg {v / n:16.1f}")
