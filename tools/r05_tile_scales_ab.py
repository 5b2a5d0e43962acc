"""Round 5: chained f16 x 2 plane products with TILE scales written by the producer's epilogue (tuning key 28 = 1, default) against
fp32 out + max|C| + split launch per chained product (key 28 = 0): time and error against fp64 (torch, on the device) of the fp32
apply with new factors, the fp32 update and the bf16-operand update.   python tools/r05_tile_scales_ab.py [apply|update|all]"""
import sys
import torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import kron  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
SHAPES = ((4096, 4096), (2048, 4096), (1000, 3000), (6144, 6144), (2304, 2048), (1100, 530), (8192, 2048))


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def ref_apply(Ql, Qr, G):
    Ql, Qr, G = Ql.double(), Qr.double(), G.double()
    return Ql.t() @ (Ql @ G @ Qr.t()) @ Qr


def ref_update(Ql, Qr, dX, dG, step):
    Ql, Qr, dX, dG = (t.double() for t in (Ql, Qr, dX, dG))
    ml, mr = Ql.diagonal().max(), Qr.diagonal().max()       # psgd.py:166-170 (max of the diagonals)
    rho = torch.sqrt(ml / mr)
    Ql, Qr = Ql / rho, Qr * rho
    A = Ql @ dG @ Qr.t()
    Bt = torch.linalg.solve_triangular(Ql.t(), torch.linalg.solve_triangular(Qr, dX, upper=True, left=False), upper=False)
    g1, g2 = torch.triu(A @ A.t() - Bt @ Bt.t()), torch.triu(A.t() @ A - Bt.t() @ Bt)
    tiny = 1.1754943508222875e-38
    return (Ql - (step / (g1.abs().max() + tiny)) * g1 @ Ql, Qr - (step / (g2.abs().max() + tiny)) * g2 @ Qr), (Ql, Qr)


g = torch.Generator(device="cuda"); g.manual_seed(3)
for M, N in SHAPES:
    Ql, Qr = tri(M, g), tri(N, g)
    G = torch.randn(M, N, device="cuda", generator=g)
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
    line = "%5d x %5d" % (M, N)
    if what in ("apply", "all"):
        want = ref_apply(Ql, Qr, G)
        for key in (0, 1):
            kron.set_tuning(28, key)
            out = psgd.precond_grad_kron(Ql.clone(), Qr.clone(), G)
            t = min(timeit(lambda: psgd.precond_grad_kron(Ql.clone(), Qr.clone(), G), 8) for _ in range(3))
            line += " | apply[%d] %.3f ms err %.2e" % (key, t, rel(out, want))
        del want
    if what in ("update", "all"):
        (wl, wr), (bl, br) = ref_update(Ql, Qr, dX, dG, 0.01)
        for key in (0, 1):
            kron.set_tuning(28, key)
            a, b = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
            t = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 6) for _ in range(3))
            tb = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX.bfloat16(), dG.bfloat16(), 0.01), 6) for _ in range(3))
            inc = max(rel(a - bl.float(), wl - bl), rel(b - br.float(), wr - br))
            line += " | update[%d] %.3f ms state %.1e incr %.1e, bf16 ops %.3f ms" % (key, t, max(rel(a, wl), rel(b, wr)), inc, tb)
        del wl, wr, bl, br
    kron.set_tuning(28, 1)
    print(line, flush=True)
    del Ql, Qr, G, dX, dG
    torch.cuda.empty_cache()
