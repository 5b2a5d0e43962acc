R=$PWD
mkdir -p gpurun_out/kupd
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kupd/stats -- python3 $R/tools/kron_big_update.py ${1:-4096} ${2:-f32} > $R/gpurun_out/kupd/out.txt 2>&1
python3 - <<PY
import sqlite3,glob
db=glob.glob('$R/gpurun_out/kupd/stats/**/*_results.db',recursive=True)[0]
con=sqlite3.connect(db)
cols=[r[1] for r in con.execute("pragma table_info(kernels)")]
name="name" if "name" in cols else "kernel_name"
scol="stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
rows=list(con.execute(f"select {name}, start, duration, {scol} from kernels order by start"))
# last update call: find last k_kron_balance
idx=[i for i,r in enumerate(rows) if 'k_kron_balance' in r[0]]
last=idx[-1]
tot=0
t0=rows[last][1]
for n,s,d,gx in rows[last:]:
    print("%-60s start %8.1f us  %9.1f us  stream/queue %s" % (n[:60], (s-t0)/1e3, d/1e3, gx)); tot+=d
end=max(s+d for n,s,d,gx in rows[last:])
print("sum of kernel times %.1f us, first start to last end %.1f us" % (tot/1e3, (end-t0)/1e3))
PY
rm -rf $R/gpurun_out/kupd/stats
