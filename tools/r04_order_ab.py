"""Round 4: stream order of the large fp32 Kron update (tuning key 25): 0 = products of :173 first on the side stream, 1 = both inversions first.
    python tools/r04_order_ab.py"""
import sys
import torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402
from tools.kron_f16_planes_ab import update_ref64, errs  # noqa: E402
lib = _lib.load()
g = torch.Generator(device="cuda"); g.manual_seed(1)
for M, N in ((4096, 4096), (2048, 4096), (3072, 3072), (6144, 6144)):
    Ql, Qr = tri(M, g), tri(N, g)
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = dX * torch.exp(torch.rand(M, 1, device="cuda", generator=g) * 2 - 1) * torch.exp(torch.rand(1, N, device="cuda", generator=g) * 2 - 1)
    rl, rr, bl, br = update_ref64(Ql, Qr, dX, dG, 0.01)
    res = {}
    for rnd in range(2):
        for key in (0, 1):
            lib.psgd_kron_set_tuning(25, key)
            t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 8)
            if rnd == 0:
                a, b = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
                res[key] = [t, errs(a, rl)[0], errs(b, rr)[0]]
            else:
                res[key][0] = min(res[key][0], t)
    print("%-10s fp32 update  order 0: %.3f ms  order 1: %.3f ms   rel %.1e/%.1e %.1e/%.1e" % ("%dx%d" % (M, N), res[0][0], res[1][0], res[0][1], res[0][2], res[1][1], res[1][2]))
lib.psgd_kron_set_tuning(25, 0)
