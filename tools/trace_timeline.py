"""Timeline of one iteration from a rocprofv3 --kernel-trace database: kernels between two launches of an anchor kernel.
    python tools/trace_timeline.py <results.db> <anchor substring> [which occurrence]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kdl = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')]
if kdl:
    kd = kdl[0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = db.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
else:                                                   # the `kernels` view of newer rocprofv3 databases
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    q = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = db.execute(f"select {name}, start, start + duration, {q} from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
i0, i1 = idx[which], idx[which + 1]
t0 = rows[i0][1]
for n, a, b, q in rows[i0:i1]:
    print('%8.1f %8.1f %7.1f q%s %s' % ((a - t0) / 1000, (b - t0) / 1000, (b - a) / 1000, q, n[:60]))
print('period %.1f us' % ((rows[i1][1] - t0) / 1000))
