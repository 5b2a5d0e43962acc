import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(0)
M, N = 1024, 1152
tri = lambda n: (np.triu(rng.standard_normal((n, n)) * 0.05, 1) + np.diag(np.exp(0.3 * rng.standard_normal(n)))).astype(np.float32)
Ql, Qr = torch.from_numpy(tri(M)).cuda(), torch.from_numpy(tri(N)).cuda()
dX, dG = torch.randn(M, N, device="cuda"), torch.randn(M, N, device="cuda")
outs = {}
for inv in (1, 0, 1):
    print("set", lib.psgd_kron_set_tuning(11, inv))
    outs[inv] = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
torch.cuda.synchronize()
print("max abs diff Ql", (outs[1][0] - outs[0][0]).abs().max().item(), "Qr", (outs[1][1] - outs[0][1]).abs().max().item())
