import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from tools.kron_timing import state
M = N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Ql, Qr, dX, dG, G = state(M, N, torch.device("cuda:0"))
for _ in range(4):
    psgd.precond_grad_kron(Ql, Qr, G)
for _ in range(2):
    psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
torch.cuda.synchronize()
