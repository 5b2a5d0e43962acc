"""Turn rocprofv3 --pmc CSV output (FETCH_SIZE / WRITE_SIZE passes) into per-launch HBM bytes per kernel.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request of a
wide coalesced stream, i.e. exactly half the bytes -> doubled here; WRITE_SIZE is exact.
rocprofv3 reports both counters in KB (x1024 bytes).
Usage: python tools/pmc_traffic.py <dir-with-counter_collection.csv> ...
"""
import collections
import csv
import glob
import json
import os
import sys


def load(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for path in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):   # rocpd output format
            import sqlite3
            con = sqlite3.connect(path)
            for name, ctr, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
                acc[name][ctr].append(float(val))
    return acc


if __name__ == "__main__":
    acc = load(sys.argv[1:])
    out = {}
    for k, c in acc.items():
        if "psgd" not in k:
            continue
        f = c.get("FETCH_SIZE")
        w = c.get("WRITE_SIZE")
        rec = {"launches": len(f or w)}
        if f:
            rec["fetch_bytes_raw"] = sum(f) / len(f) * 1024
            rec["fetch_bytes_corrected_x2"] = 2 * rec["fetch_bytes_raw"]
        if w:
            rec["write_bytes"] = sum(w) / len(w) * 1024
        if f and w:
            rec["hbm_bytes_per_launch"] = rec["fetch_bytes_corrected_x2"] + rec["write_bytes"]
        out[k.split("(")[0]] = rec
    print(json.dumps(out, indent=1))
