"""The solves of the large fp32 Kron update: substitution strips (tuning key 11 = 0) against explicit inverses on f16 x 2 planes
(key 11 = 1): time, error of the new factors and of their increments against an fp64 run.
    python tools/trsm_inv_ab.py
"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402
from tools.kron_f16_planes_ab import update_ref64, errs  # noqa: E402

if __name__ == "__main__":
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    shapes = ((4096, 4096), (2048, 2048), (1024, 1024), (1300, 1100), (3000, 2048), (2048, 4096), (8192, 1024), (2944, 2944), (6144, 6144))
    if len(sys.argv) > 1 and sys.argv[1] == "mid":
        shapes = ((1024, 1024), (1280, 1280), (1536, 1536), (1792, 1792), (2048, 2048), (1300, 1100), (1024, 4096), (1536, 3072), (2048, 1024))
    for M, N in shapes:
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * torch.exp(torch.rand(M, 1, device="cuda", generator=g) * 2 - 1) * torch.exp(torch.rand(1, N, device="cuda", generator=g) * 2 - 1)
        rl, rr, bl, br = update_ref64(Ql, Qr, dX, dG, 0.01)
        res = []
        for rnd in range(2):
            for i, key in enumerate((0, 1)):
                lib.psgd_kron_set_tuning(11, key)
                t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 8)
                if rnd == 0:
                    a, b = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
                    a2, b2 = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
                    il = ((a.double() - bl) - (rl - bl)).norm() / (rl - bl).norm()
                    ir = ((b.double() - br) - (rr - br)).norm() / (rr - br).norm()
                    res.append([t, errs(a, rl)[0], errs(b, rr)[0], il.item(), ir.item(), torch.equal(a, a2) and torch.equal(b, b2)])
                else:
                    res[i][0] = min(res[i][0], t)
        a, b = res
        print("%-10s update %.3f -> %.3f ms (%+.0f%%)  rel %.1e/%.1e -> %.1e/%.1e  increment %.1e/%.1e -> %.1e/%.1e  rep %s" %
              ("%dx%d" % (M, N), a[0], b[0], (b[0] / a[0] - 1) * 100, a[1], a[2], b[1], b[2], a[3], a[4], b[3], b[4], b[5]))
    # bf16 operands for the products, fp32 solves (psgd_kron_dd_update_bf16): the same two routes for the solves
    for M, N in ((4096, 4096), (2304, 2048), (2048, 4096), (6144, 6144)):
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g).bfloat16()
        dG = (dX.float() * 1.5).bfloat16()
        rl, rr, bl, br = update_ref64(Ql, Qr, dX.float(), dG.float(), 0.01)
        res = []
        for rnd in range(2):
            for i, key in enumerate((0, 1)):
                lib.psgd_kron_set_tuning(11, key)
                t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 8)
                if rnd == 0:
                    a, b = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
                    il = ((a.double() - bl) - (rl - bl)).norm() / (rl - bl).norm()
                    ir = ((b.double() - br) - (rr - br)).norm() / (rr - br).norm()
                    res.append([t, il.item(), ir.item()])
                else:
                    res[i][0] = min(res[i][0], t)
        a, b = res
        print("%-10s bf16-operand update %.3f -> %.3f ms (%+.0f%%)  increment error %.1e/%.1e -> %.1e/%.1e" %
              ("%dx%d" % (M, N), a[0], b[0], (b[0] / a[0] - 1) * 100, a[1], a[2], b[1], b[2]))
    lib.psgd_kron_set_tuning(11, 1)
