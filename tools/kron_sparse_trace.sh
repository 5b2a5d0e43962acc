R=$PWD
mkdir -p gpurun_out/spt
cat > /tmp/sp.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["R"])
import preconditioned_stochastic_gradient_descent as psgd
g = torch.Generator(device="cuda"); g.manual_seed(0)
def fac(kind, n):
    if kind == "dense":
        return torch.triu(torch.randn(n, n, device="cuda", generator=g) * 0.02, 1) + torch.eye(n, device="cuda")
    if kind == "norm":
        q = torch.stack([torch.exp(0.2 * torch.randn(n, device="cuda", generator=g)), 0.1 * torch.randn(n, device="cuda", generator=g)]); q[1, -1] = 0.0
        return q
    return torch.exp(0.2 * torch.randn(1, n, device="cuda", generator=g))
for kl, kr, M, N in (("norm", "scale", 30000, 1000), ("dense", "norm", 1000, 30000), ("norm", "dense", 30000, 1000)):
    Ql, Qr = fac(kl, M), fac(kr, N)
    dX = torch.randn(M, N, device="cuda", generator=g); dG = dX * 1.5; G = torch.randn(M, N, device="cuda", generator=g)
    for _ in range(2):
        psgd.precond_grad_kron(Ql, Qr, G)
    for _ in range(2):
        psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
R=$R rocprofv3 --kernel-trace --stats -d $R/gpurun_out/spt/stats -- python3 /tmp/sp.py > /dev/null 2>&1
python3 - <<PY
import sqlite3,glob,collections
db=glob.glob('$R/gpurun_out/spt/stats/**/*_results.db',recursive=True)[0]
con=sqlite3.connect(db)
cols=[r[1] for r in con.execute("pragma table_info(kernels)")]
name="name" if "name" in cols else "kernel_name"
acc=collections.OrderedDict()
for n,d in con.execute(f"select {name}, duration from kernels order by start"):
    if 'psgdk' in n or 'elementwise' in n or 'copy' in n.lower():
        k=n.split('(')[0][:60]; acc.setdefault(k,[0,0.0]); acc[k][0]+=1; acc[k][1]+=d/1e3
for k,(c,t) in sorted(acc.items(), key=lambda x:-x[1][1])[:16]: print("%-62s x%-3d %9.1f us  (%.1f each)"%(k,c,t,t/c))
PY
rm -rf $R/gpurun_out/spt
