#!/usr/bin/env python3
"""Per-kernel summary of the LAST training step in a rocprofv3 --kernel-trace database of tools/train_bench.py
(the replayed HIP graph: one step = the kernels after the previous step's last weight-gradient kernel)."""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'") if 'kernel_dispatch' in r[0]][0]
sfx = t.split('rocpd_kernel_dispatch_')[1]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y from rocpd_kernel_dispatch_{sfx} d "
                   f"join rocpd_info_kernel_symbol_{sfx} s on d.kernel_id=s.id order by d.start").fetchall()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
idx = [i for i, r in enumerate(rows) if 'conv_wgrad16' in r[0] or 'conv_wgrad_kernel' in r[0]]
per = len(idx) // steps
win = rows[idx[-per - 1] + 1:]
short = lambda n: re.sub(r'_ZN\d+_GLOBAL__N_1\d+', '', n)[:56]
print(f"{len(win)} kernels in the last step; span {(win[-1][2] - win[0][1]) / 1e6:.2f} ms, sum of durations {sum(r[2] - r[1] for r in win) / 1e6:.2f} ms")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    a = agg[short(r[0])]; a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
print("Name,Calls,TotalUs,AverageUs")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"\"{k}\",{v[0]},{v[1]:.1f},{v[1] / v[0]:.2f}")
if len(sys.argv) > 3:
    for r in sorted(win, key=lambda r: -(r[2] - r[1]))[:int(sys.argv[3])]:
        print(f"# {(r[2] - r[1]) / 1e3:8.0f} us  grid {r[3] // 256}x{r[4]}  {short(r[0])}")
