"""One sparse-format shape in a loop for `rocprofv3 --kernel-trace`:   python tools/kron_sparse_trace.py kind_l kind_r M N [apply|update]"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from tools.kron_bf16_update_timing import tri  # noqa: E402

if __name__ == "__main__":
    kl, kr, M, N = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    what = sys.argv[5] if len(sys.argv) > 5 else "update"
    g = torch.Generator(device="cuda"); g.manual_seed(0)

    def fac(kind, n):
        if kind == "dense":
            return tri(n, g)
        if kind == "norm":
            q = torch.stack([torch.exp(0.2 * torch.randn(n, device="cuda", generator=g)), 0.1 * torch.randn(n, device="cuda", generator=g)])
            q[1, -1] = 0.0
            return q
        return torch.exp(0.2 * torch.randn(1, n, device="cuda", generator=g))
    Ql, Qr = fac(kl, M), fac(kr, N)
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
    for _ in range(12):
        if what == "apply":
            psgd.precond_grad_kron(Ql, Qr, dX)
        else:
            psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
    torch.cuda.synchronize()
