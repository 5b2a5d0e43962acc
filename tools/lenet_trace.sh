# per-kernel trace of the batched LeNet5-set Kron apply and update (rocprofv3 kernel trace)
R=$PWD
mkdir -p gpurun_out/lenet
cat > /tmp/lenet.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["R"])
import preconditioned_stochastic_gradient_descent as psgd
from tools.kron_timing import state
dev = torch.device("cuda:0")
LENET5 = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10)]
sts = [state(m, n, dev) for m, n in LENET5]
Qls, Qrs, dXs, dGs, Gs = ([s[i] for s in sts] for i in range(5))
for _ in range(5):
    psgd.precond_grad_kron_batched(Qls, Qrs, Gs)
torch.cuda.synchronize()
for _ in range(3):
    psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
R=$R rocprofv3 --kernel-trace --stats -d $R/gpurun_out/lenet/stats -- python3 /tmp/lenet.py > /dev/null 2>&1
python3 - <<PY
import sqlite3,glob
db=glob.glob('$R/gpurun_out/lenet/stats/**/*_results.db',recursive=True)[0]
con=sqlite3.connect(db)
cols=[r[1] for r in con.execute("pragma table_info(kernels)")]
name="name" if "name" in cols else "kernel_name"
rows=list(con.execute(f"select {name}, start, end from kernels order by start"))
rows=[r for r in rows if 'psgdk' in r[0]]
last=rows[-60:]
prev=None
for n,s,e in last:
    print("%-58s %7.1f us   gap %6.1f us" % (n[:58], (e-s)/1e3, (s-prev)/1e3 if prev else 0)); prev=e
PY
rm -rf $R/gpurun_out/lenet/stats
