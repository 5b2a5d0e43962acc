# per-kernel trace of one fp32 Kron apply at 4096^2 (rocprofv3 kernel trace)
R=$PWD
mkdir -p gpurun_out/kapp
cat > /tmp/kapp.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["R"])
import preconditioned_stochastic_gradient_descent as psgd
from tools.kron_timing import state
M = N = 4096
Ql, Qr, dX, dG, G = state(M, N, torch.device("cuda:0"))
Ql2 = Ql.clone()
for _ in range(3):
    Ql.add_(0.0)
    psgd.precond_grad_kron(Ql, Qr, G)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
R=$R rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kapp/stats -- python3 /tmp/kapp.py > /dev/null 2>&1
python3 - <<PY
import sqlite3,glob
db=glob.glob('$R/gpurun_out/kapp/stats/**/*_results.db',recursive=True)[0]
con=sqlite3.connect(db)
cols=[r[1] for r in con.execute("pragma table_info(kernels)")]
name="name" if "name" in cols else "kernel_name"
rows=list(con.execute(f"select {name}, start, duration from kernels order by start"))
for n,s,d in rows[-6:]: print("%-60s %9.1f us" % (n[:60], d/1e3))
PY
rm -rf $R/gpurun_out/kapp
