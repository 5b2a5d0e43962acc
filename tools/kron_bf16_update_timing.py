"""Times the bf16-operand Kron update against the fp32 one (run on the GPU box).
    python tools/kron_bf16_update_timing.py [M N]"""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402


def tri(n, g):
    return torch.triu(torch.randn(n, n, device="cuda", generator=g) * 0.02, 1) + torch.diag(torch.exp(0.3 * torch.randn(n, device="cuda", generator=g)))


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


if __name__ == "__main__":
    M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    Ql, Qr = tri(M, g), tri(N, g)
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
    dXb, dGb = dX.bfloat16(), dG.bfloat16()
    t32 = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dXb.float(), dGb.float(), 0.01))
    tb = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01))
    a, b = psgd.update_precond_kron(Ql, Qr, dXb.float(), dGb.float(), 0.01), psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01)
    err = max(((x - y).norm() / y.norm()).item() for x, y in zip(b, a))
    print("Kron update %dx%d: fp32 %.3f ms (incl. bf16->fp32 casts), bf16 operands %.3f ms, max rel diff of factors %.2e" % (M, N, t32, tb, err))
