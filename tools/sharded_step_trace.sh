# Kernel timeline of the fused UVd step at one rank's share of BASELINE configs[3] (12.5M rows, r = 20), through the
# multi-GPU path on a 1-rank RCCL group and unsharded: which kernels and gaps the two exchanges add.
#   bash tools/sharded_step_trace.sh        (GPU box; output gpurun_out/shtrace/{sharded,unsharded}.txt)
R=$PWD
mkdir -p gpurun_out/shtrace
cd /tmp && export TMPDIR=/tmp
for mode in sharded unsharded; do
  FLAG=""; [ $mode = sharded ] && FLAG="--force-sharded"
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/shtrace/$mode -- python3 $R/bench.py --gpus 1 --rows 12500032 $FLAG --no-kron --no-legs --no-cpu-baseline --no-exchange-leg --steps 30 --warmup 5 > $R/gpurun_out/shtrace/$mode.log 2>&1
  python3 - <<PY > $R/gpurun_out/shtrace/$mode.txt
import sqlite3, glob, re
from collections import defaultdict
db = glob.glob('$R/gpurun_out/shtrace/$mode/**/*_results.db', recursive=True)[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = [(s, s + d, re.sub(r"\(.*", "", n).replace("void psgd::", "").replace("psgd::", "")) for n, s, d in
        con.execute(f"select {name}, start, duration from kernels order by start")]
steps, cur = [], []
for s, e, n in rows:
    if n.startswith("k_update_gram") and cur:
        steps.append(cur); cur = []
    cur.append((s, e, n))
steps = [st for st in steps[8:30] if len(st) == len(steps[10])]
acc = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for st in steps:
    t0 = st[0][0]; prev = None
    for i, (s, e, n) in enumerate(st):
        a = acc[(i, n[:44])]
        a[0] += s - t0; a[1] += e - s; a[2] += (s - prev) if prev is not None else 0; a[3] += 1; prev = e
print("$mode: fused UVd step, 12.5M rows, r = 20; averages over %d steps" % len(steps))
for (i, n), a in sorted(acc.items()):
    print("%2d %-44s start %8.1f us  dur %7.1f us  gap before %6.1f us" % (i, n, a[0] / a[3] / 1e3, a[1] / a[3] / 1e3, a[2] / a[3] / 1e3))
per = [(steps[i + 1][0][0] - steps[i][0][0]) / 1e3 for i in range(len(steps) - 1)]
print("step period %.1f us" % (sum(per) / max(len(per), 1)))
PY
  rm -rf $R/gpurun_out/shtrace/$mode
done
