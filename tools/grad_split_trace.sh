R=$PWD
mkdir -p gpurun_out/gst
cat > /tmp/gs.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["R"])
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import _lib
from tools.kron_timing import state
lib = _lib.load()
M = N = 4096
Ql, Qr, dX, dG, G = state(M, N, torch.device("cuda:0"))
for split in (0, 1, 0, 1):
    lib.psgd_kron_set_tuning(6, split)
    for _ in range(3):
        psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
R=$R rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gst/stats -- python3 /tmp/gs.py > /dev/null 2>&1
python3 - <<PY
import sqlite3,glob
db=glob.glob('$R/gpurun_out/gst/stats/**/*_results.db',recursive=True)[0]
con=sqlite3.connect(db)
cols=[r[1] for r in con.execute("pragma table_info(kernels)")]
name="name" if "name" in cols else "kernel_name"
rows=list(con.execute(f"select {name}, start, duration from kernels order by start"))
for n,s,d in rows:
    if 'p3_grad' in n or 'p3_pair' in n: print("%-50s %9.1f us" % (n[:50], d/1e3))
PY
rm -rf $R/gpurun_out/gst
