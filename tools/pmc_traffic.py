#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv > profiles/pmc_traffic.json

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE reports half the bytes of
16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  Kernel names are
mapped to the family names bench.py prints."""
import csv, json, re, sys, collections

def family(name):
    dt = lambda t: "f16" if "Float16" in t else "bf16"
    m = re.search(r"igemm2_kernel<(unsigned short|_Float16), (\d+), (\d+), \d+, \d+, (\d+), (true|false)(?:, (\d+))?>", name)
    if m: return f"igemm2<{dt(m.group(1))},{m.group(2)}x{m.group(3)},s{m.group(4)}" + (",kg2>" if m.group(6) == "2" else ">")
    if "conv3x3_halo_kernel" in name: return "conv3x3_halo<" + dt(name) + ">"
    m = re.search(r"conv_igemm(?:_splitk)?_kernel<(unsigned short|_Float16), (\d+), (\d+)", name)
    if m: return f"conv_igemm<{dt(m.group(1))},{m.group(2)}x{m.group(3)}>"
    m = re.search(r"conv3x3_direct_kernel<(unsigned short|_Float16), (\d+), (\d+)", name)
    if m: return f"conv3x3_direct<{dt(m.group(1))},{m.group(2)}x16px,{m.group(3)}>"
    m = re.search(r"(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+_kernel)", name)
    return m.group(1) if m else name[:60]

agg = collections.defaultdict(lambda: {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            c = r["Counter_Name"]
            if c not in ("FETCH_SIZE", "WRITE_SIZE"): continue
            e = agg[family(r["Kernel_Name"])][c]
            e[0] += 1; e[1] += float(r["Counter_Value"])
out = {}
for k, v in sorted(agg.items()):
    if "at::native" in k or not v["FETCH_SIZE"][0] or not v["WRITE_SIZE"][0]: continue
    f_kib = v["FETCH_SIZE"][1] / v["FETCH_SIZE"][0]
    w_kib = v["WRITE_SIZE"][1] / v["WRITE_SIZE"][0]
    out[k] = {"launches_fetch_pass": v["FETCH_SIZE"][0], "launches_write_pass": v["WRITE_SIZE"][0],
              "FETCH_SIZE_KiB_avg": round(f_kib, 1), "WRITE_SIZE_KiB_avg": round(w_kib, 1),
              "hbm_bytes_per_launch": round((2.0 * f_kib + w_kib) * 1024.0)}
json.dump(out, sys.stdout, indent=1)
