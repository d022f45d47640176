#!/usr/bin/env python3
"""Which torch (at::native) kernels run INSIDE the captured training step, and between which of our kernels: from the rocprofv3
--kernel-trace database of `tools/train_bench.py --graph` (last replayed step).  Found the five-launch torch construction of the
Toeplitz band tables (now cfp_dwconv_large_toeplitz).

    rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp -- python3 tools/train_bench.py --dtype bf16 --graph --steps 8
    python3 tools/torch_kernels_in_step.py /tmp/tp/*.db
"""
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'") if 'kernel_dispatch' in r[0]][0]
sfx = t.split('rocpd_kernel_dispatch_')[1]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from rocpd_kernel_dispatch_{sfx} d join rocpd_info_kernel_symbol_{sfx} s on d.kernel_id=s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if 'conv_wgrad16' in r[0]]
per = len(idx) // 12
win = rows[idx[-per - 1] + 1:]
agg = collections.Counter(); tt = collections.Counter()
prev = None
ctx = collections.defaultdict(collections.Counter)
for i, r in enumerate(win):
    if 'at::native' in r[0] or '_ZN2at' in r[0]:
        n = re.sub(r'\s+', ' ', r[0])[:200]
        agg[(n, r[3])] += 1; tt[(n, r[3])] += (r[2]-r[1])/1e3
        ctx[(n, r[3])][(win[i-1][0][:50] if i else '', win[i+1][0][:50] if i+1 < len(win) else '')] += 1
for k, c in agg.most_common(20):
    print(c, round(tt[k],1), k[1], k[0][:160])
    for (a, b), cc in ctx[k].most_common(2): print('     between', a, '|', b, cc)
