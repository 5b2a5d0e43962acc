import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from tools.kron_timing import state
dev = torch.device("cuda:0")
for (M, N) in [(64, 64), (256, 64), (1024, 64), (2048, 64), (257, 120)]:
    Ql, Qr, dX, dG, G = state(M, N, dev)
    for _ in range(3):
        psgd.precond_grad_kron(Ql, Qr, G)
    torch.cuda.synchronize()
