"""Scan of shapes for performance cliffs: fp32 apply / update and bf16-operand apply over a grid of (M, N), reported as
F_ref TFLOP/s (the dense flops of psgd.py:189-192 / :173-179), slowest first.   python tools/kron_shape_scan.py"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402
from tools.kron_timing import flops_apply, flops_update  # noqa: E402

if __name__ == "__main__":
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    dims = [200, 384, 500, 700, 1000, 1300, 1700, 2048, 2500, 3072]
    rows = []
    for M in dims:
        for N in dims:
            if M * N < 300 * 700:
                continue
            Ql, Qr = tri(M, g), tri(N, g)
            dX = torch.randn(M, N, device="cuda", generator=g)
            dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
            G = torch.randn(M, N, device="cuda", generator=g)
            Gb = G.bfloat16()
            ta = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), 10)
            tu = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 5)
            tb = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, Gb), 10)
            rows.append((M, N, ta, tu, tb, flops_apply(M, N) / ta * 1e-9, flops_update(M, N) / tu * 1e-9, flops_apply(M, N) / tb * 1e-9))
    print("   M     N   apply ms  update ms  bf16 apply ms | TFLOP/s (F_ref): apply  update  bf16 apply")
    for r in rows:
        print("%5d %5d   %8.3f  %9.3f  %13.3f |                  %6.1f  %6.1f  %10.1f" % r)
