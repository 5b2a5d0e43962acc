"""A/B of the two plane formats of the large fp32 apply (tuning key 12: 0 = bf16 x 3, 1 = f16 x 2): time and norm-wise /
element-wise error against an fp64 product, over shapes, magnitudes of G and dynamic ranges of the factors.

    python tools/kron_f16_planes_ab.py
"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib, kron  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402


def wide_tri(n, g, decades):
    """upper-triangular factor whose diagonal spans `decades` powers of ten"""
    q = torch.triu(torch.randn(n, n, device="cuda", generator=g) * 0.02, 1)
    d = torch.exp(torch.linspace(-1.15 * decades, 1.15 * decades, n, device="cuda"))
    return q * d[:, None] + torch.diag(d[torch.randperm(n, device="cuda", generator=g)])


def errs(out, ref):
    d = (out.double() - ref)
    return (d.norm() / ref.norm()).item(), (d.abs().max() / ref.abs().max()).item()


def run(lib, Ql, Qr, G, label, reps=30):
    ref = (Ql.double().T @ Ql.double()) @ G.double() @ (Qr.double().T @ Qr.double())
    res = []
    for f16 in (0, 1):
        lib.psgd_kron_set_tuning(12, f16)
        kron.invalidate_factor_cache()
        old = kron.set_factor_cache(False)
        cold = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), reps)
        kron.set_factor_cache(True)
        warm = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), reps)
        out = psgd.precond_grad_kron(Ql, Qr, G)
        out2 = psgd.precond_grad_kron(Ql, Qr, G)
        kron.set_factor_cache(old)
        res.append((cold, warm) + errs(out, ref) + (torch.equal(out, out2), bool(torch.isfinite(out).all())))
    a, b = res
    print("%-34s cold %.3f -> %.3f ms  cached %.3f -> %.3f ms (%+.0f%%)  rel %.1e -> %.1e  max-rel %.1e -> %.1e  rep %s fin %s" %
          (label, a[0], b[0], a[1], b[1], (b[1] / a[1] - 1) * 100, a[2], b[2], a[3], b[3], b[4], b[5]))


def update_ref64(Ql, Qr, dX, dG, step):
    """psgd.py:156-179 in fp64 on the device"""
    Ql, Qr, dX, dG = (t.double() for t in (Ql, Qr, dX, dG))
    rho = torch.sqrt(Ql.diagonal().max() / Qr.diagonal().max())
    Ql, Qr = Ql / rho, Qr * rho
    A = Ql @ (dG @ Qr.T)
    X1 = torch.linalg.solve_triangular(Qr, dX, upper=True, left=False)               # dX Qr^-1
    Bt = torch.linalg.solve_triangular(Ql.T, X1, upper=False, left=True)             # Ql^-T (.)
    g1, g2 = torch.triu(A @ A.T - Bt @ Bt.T), torch.triu(A.T @ A - Bt.T @ Bt)
    tiny = torch.finfo(torch.float32).tiny
    return Ql - (step / (g1.abs().max() + tiny)) * g1 @ Ql, Qr - (step / (g2.abs().max() + tiny)) * g2 @ Qr, Ql, Qr


def run_update(lib, M, N, g, label, reps=10):
    Ql, Qr = tri(M, g), tri(N, g)
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = dX * torch.exp(torch.rand(M, 1, device="cuda", generator=g) * 2 - 1) * torch.exp(torch.rand(1, N, device="cuda", generator=g) * 2 - 1)
    rl, rr, bl, br = update_ref64(Ql, Qr, dX, dG, 0.01)
    res = []
    for f16 in (0, 2):
        lib.psgd_kron_set_tuning(12, f16)
        t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), reps)
        a, b = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
        a2, b2 = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
        # error of the INCREMENT (new - balanced old), relative to its norm: what the step size multiplies
        il = ((a.double() - bl) - (rl - bl)).norm() / (rl - bl).norm()
        ir = ((b.double() - br) - (rr - br)).norm() / (rr - br).norm()
        res.append((t, errs(a, rl)[0], errs(b, rr)[0], il.item(), ir.item(), torch.equal(a, a2) and torch.equal(b, b2)))
    a, b = res
    print("%-14s update %.3f -> %.3f ms (%+.0f%%)  rel %.1e/%.1e -> %.1e/%.1e  increment %.1e/%.1e -> %.1e/%.1e  rep %s" %
          (label, a[0], b[0], (b[0] / a[0] - 1) * 100, a[1], a[2], b[1], b[2], a[3], a[4], b[3], b[4], b[5]))


if __name__ == "__main__":
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for M, N in ((4096, 4096), (2048, 4096), (4096, 1024), (2944, 2944), (1000, 1000), (1024, 1024), (520, 3000), (500, 500),
                 (8192, 1024)):
        run_update(lib, M, N, g, "%dx%d" % (M, N))
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for M, N in ((4096, 4096), (2048, 4096), (4096, 1024), (1000, 1000), (1024, 1024), (520, 3000), (300, 4000), (128, 4096),
                 (64, 8192), (2048, 512), (700, 700)):
        run(lib, tri(M, g), tri(N, g), torch.randn(M, N, device="cuda", generator=g), "%dx%d" % (M, N))
    M = N = 2048
    for scale in (1e-37, 1e-30, 1e-12, 1e-4, 1e6, 1e20):
        run(lib, tri(M, g), tri(N, g), torch.randn(M, N, device="cuda", generator=g) * scale, "2048^2 G x %.0e" % scale, 5)
    for dec in (1, 2, 3):
        run(lib, wide_tri(M, g, dec), wide_tri(N, g, dec), torch.randn(M, N, device="cuda", generator=g),
            "2048^2 factors over 10^+-%d" % dec, 5)
    # rows of G of very different magnitude (element-wise dynamic range inside one matrix)
    G = torch.randn(M, N, device="cuda", generator=g) * torch.exp(torch.linspace(-14, 14, M, device="cuda"))[:, None]
    run(lib, tri(M, g), tri(N, g), G, "2048^2 rows of G over e^+-14", 5)
    G = torch.zeros(M, N, device="cuda"); G[5, 7] = 3.0
    run(lib, tri(M, g), tri(N, g), G, "2048^2 single non-zero", 5)
    lib.psgd_kron_set_tuning(12, 1)
