"""One shape's fp32 Kron apply with NEW factors on every call, in a loop, for `rocprofv3 --kernel-trace`:
    python tools/kron_apply_trace.py M N [reps] [route]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import kron  # noqa: E402

if __name__ == "__main__":
    M, N = int(sys.argv[1]), int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    if len(sys.argv) > 4:
        kron.set_apply_route(sys.argv[4])
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda n: torch.triu(torch.randn(n, n, device=dev, generator=g) * 0.02, 1) + torch.eye(n, device=dev)
    pairs = [(mk(M), mk(N)), (mk(M), mk(N))]
    G = torch.randn(M, N, device=dev, generator=g)
    for i in range(reps):
        psgd.precond_grad_kron(pairs[i & 1][0], pairs[i & 1][1], G)
    torch.cuda.synchronize()
