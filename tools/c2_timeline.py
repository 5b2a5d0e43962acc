"""Timeline of one fused UVd step at config 2 from a rocprofv3 kernel trace (rocpd db): per kernel start offset,
duration and the gap to its predecessor, averaged over the steady-state steps.   python tools/c2_timeline.py <db>"""
import re
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = [(s, e, re.sub(r"\(.*", "", n).replace("void psgd::", "").replace("psgd::", "")) for n, s, e in
        con.execute(f"select {name}, start, end from kernels order by start") if "psgd" in n]
# split into steps at every k_update_gram
steps, cur = [], []
for s, e, n in rows:
    if n.startswith("k_update_gram") and cur:
        steps.append(cur)
        cur = []
    cur.append((s, e, n))
steps = [st for st in steps[20:] if len(st) == len(steps[25])]
acc = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for st in steps:
    t0 = st[0][0]
    prev_end = None
    for i, (s, e, n) in enumerate(st):
        a = acc[(i, n.split("<")[0])]
        a[0] += s - t0
        a[1] += e - s
        a[2] += (s - prev_end) if prev_end is not None else 0
        a[3] += 1
        prev_end = e
tot = 0
for (i, n), a in sorted(acc.items()):
    print("%d %-22s start %7.1f us  dur %6.1f us  gap before %5.1f us" % (i, n, a[0] / a[3] / 1e3, a[1] / a[3] / 1e3, a[2] / a[3] / 1e3))
if steps:
    per = [(steps[i + 1][0][0] - steps[i][0][0]) / 1e3 for i in range(len(steps) - 1)]
    print("step period %.1f us over %d steps" % (sum(per) / max(len(per), 1), len(per)))
