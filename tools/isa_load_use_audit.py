"""Lists kernels whose ISA has many `global_load; s_waitcnt vmcnt(0)` pairs close together (a load whose value is used at
once: one memory latency each when they sit in an unrolled loop).  Reads the .s files hipcc --save-temps leaves.
    python tools/isa_load_use_audit.py /tmp/dir/*.s"""
import re
import sys

for path in sys.argv[1:]:
    name, loads, hits, n = None, 0, 0, 0
    out = []
    last_load = -100
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                out.append((hits, loads, name))
            name, loads, hits, n, last_load = m.group(1), 0, 0, 0, -100
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        n += 1
        if t.startswith("global_load") or t.startswith("buffer_load"):
            loads += 1
            last_load = n
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t and n - last_load <= 3:
            hits += 1
    if name:
        out.append((hits, loads, name))
    for hits, loads, name in sorted(out, reverse=True)[:12]:
        if hits >= 6:
            print("%4d load-then-wait pairs of %4d loads  %s" % (hits, loads, name[:110]))
