import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
dev = torch.device("cuda:0")
for r in (4, 10, 20, 32):
    N = 4096
    U = torch.randn(N, r, device=dev) * 0.01; V = torch.randn(N, r, device=dev) * 0.01
    d = torch.ones(N, 1, device=dev); v = torch.randn(N, 1, device=dev); h = torch.randn(N, 1, device=dev)
    for i in range(6):
        psgd.update_precond_UVd_math_(U, V, d, v, h, 0.01, 1e-38, balance=False, update_U=True)
    torch.cuda.synchronize()
