"""Timing of the sparse Kron formats (psgd.py:198-391) at embedding-like sizes against dense(x)dense of the same dense side.
    python tools/kron_sparse_scan.py"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    g = torch.Generator(device="cuda"); g.manual_seed(0)

    def fac(kind, n):
        if kind == "dense":
            return tri(n, g)
        if kind == "norm":
            q = torch.stack([torch.exp(0.2 * torch.randn(n, device="cuda", generator=g)), 0.1 * torch.randn(n, device="cuda", generator=g)])
            q[1, -1] = 0.0
            return q
        return torch.exp(0.2 * torch.randn(1, n, device="cuda", generator=g))

    for kl, kr, M, N in (("dense", "scale", 2048, 2048), ("dense", "norm", 2048, 2048), ("scale", "dense", 2048, 2048), ("norm", "dense", 2048, 2048),
                         ("dense", "scale", 4096, 512), ("dense", "norm", 1000, 30000), ("norm", "dense", 30000, 1000), ("dense", "scale", 1000, 30000),
                         ("norm", "scale", 30000, 1000), ("dense", "dense", 2048, 2048)):
        Ql, Qr = fac(kl, M), fac(kr, N)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
        G = torch.randn(M, N, device="cuda", generator=g)
        ta = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), 10)
        tu = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 10)
        print("%-6s (x) %-6s %6d x %-6d apply %.3f ms   update %.3f ms" % (kl, kr, M, N, ta, tu))
