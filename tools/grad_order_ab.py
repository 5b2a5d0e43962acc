"""Tile order of the gradient grid of the large fp32 Kron update (tuning key 17).   python tools/grad_order_ab.py"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for M, N in ((4096, 4096), (2944, 2944), (2048, 4096), (1024, 1024), (6144, 6144)):
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5
        res = {}
        base = None
        for rnd in range(3):
            for key in (0, 1, 2):
                lib.psgd_kron_set_tuning(17, key)
                t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 8)
                res[key] = min(res.get(key, 1e9), t)
                if rnd == 0:
                    out = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
                    if base is None:
                        base = out
                    d = max(((a - b).norm() / b.norm()).item() for a, b in zip(out, base))
                    assert d < 1e-6, d           # (which tiles the K-split tail takes depends on the order: fp32 rounding)
        print("%dx%d update: order 0 %.3f ms   1 %.3f ms   2 %.3f ms" % (M, N, res[0], res[1], res[2]))
    lib.psgd_kron_set_tuning(17, 0)
