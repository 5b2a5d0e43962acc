"""A/B of the K-split tail of the gradient grid (tuning key 6) in the fp32 Kron update; results must agree to fp32 rounding.
    python tools/grad_split_ab.py [M N]"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    shapes = [(4096, 4096), (2048, 2048), (3000, 5000), (1024, 1024)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for M, N in shapes:
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
        outs = {}
        for split in (0, 1, 0, 1):
            lib.psgd_kron_set_tuning(6, split)
            t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 10)
            a = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
            b = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
            same = all(torch.equal(x, y) for x, y in zip(a, b))
            outs[split] = a
            print("Kron fp32 update %dx%d split=%d: %.3f ms, repeatable %s" % (M, N, split, t, same))
        d = max(((x - y).norm() / y.norm()).item() for x, y in zip(outs[1], outs[0]))
        print("      rel diff of the factors split vs whole: %.1e" % d)
    lib.psgd_kron_set_tuning(6, 1)
