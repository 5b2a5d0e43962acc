"""Per-kernel duration summary (the table `rocprofv3 --stats` prints) from a rocpd `*_results.db`.

rocprofv3 on this image writes the rocpd SQLite format unless --output-format csv is given; this
turns its `kernels` view into the same columns as `*_kernel_stats.csv`.
Usage: python tools/rocpd_stats.py <dir-or-db> > profiles/<name>.csv
"""
import glob
import os
import sqlite3
import statistics
import sys
from collections import defaultdict


def main(path):
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True)
    dur = defaultdict(list)
    for db in dbs:
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        name = "name" if "name" in cols else "kernel_name"
        for n, d in con.execute(f"select {name}, duration from kernels"):
            dur[n].append(int(d))
    total = sum(sum(v) for v in dur.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        sd = statistics.pstdev(v) if len(v) > 1 else 0.0
        n = n if len(n) < 160 else n[:157] + "..."
        print(f'"{n}",{len(v)},{sum(v)},{sum(v) / len(v):.6f},{100.0 * sum(v) / total:.2f},{min(v)},{max(v)},{sd:.6f}')


if __name__ == "__main__":
    main(sys.argv[1])
