#!/usr/bin/env python3
"""Per-kernel-family summary of one rocprofv3 --pmc pass over the SQ counters (MFMA busy, wave cycles, waits, LDS conflicts).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d DIR -o q --output-format csv -- python3 bench.py ...
    python tools/pmc_sq_summary.py DIR/q_counter_collection.csv > profiles/pmc_sq_<what>.json

Per family (summed over its launches): raw counter sums, and
  mfma_busy_of_wave_cycles = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES)   (WAVE_CYCLES counts quad-cycles, MI355X_MICROARCH.md)
  wait_frac / issue_stall_frac / active_frac = WAIT_ANY, WAIT_INST_ANY, ACTIVE_INST_ANY over WAVE_CYCLES (disjoint buckets)
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
The first is "of the cycles waves were resident, the share in which a matrix instruction was executing for them" -- a per-wave
view, not chip utilisation (a kernel that fills a quarter of the chip is not penalised)."""
import collections, csv, json, re, sys


def family(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:80]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = family(r["Kernel_Name"])
            if k.startswith("at::") or "elementwise" in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
out = {}
for k, c in agg.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    e = {"launches": len(calls[k]), **{n: v for n, v in sorted(c.items())}}
    if wc > 0:
        e["mfma_busy_of_wave_cycles"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * wc), 4)
        e["wait_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4)
        e["issue_stall_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
        e["active_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)
    if c.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
        e["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    out[k] = e
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))), sys.stdout, indent=1)
