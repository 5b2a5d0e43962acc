"""A/B of split-K for few-tile plane products in the fp32 apply (tuning key 8).   python tools/kron_splitk_ab.py"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for M, N in ((1000, 1000), (1024, 1024), (700, 700), (500, 1700), (384, 2500), (200, 3072), (130, 5000), (64, 8192), (300, 4000),
                 (520, 3000), (900, 1400), (1024, 2048), (1300, 1300), (2048, 512)):
        Ql, Qr = tri(M, g), tri(N, g)
        G = torch.randn(M, N, device="cuda", generator=g)
        ref = (Ql.double().T @ Ql.double()) @ G.double() @ (Qr.double().T @ Qr.double())
        res = []
        for sk in (0, 1):
            lib.psgd_kron_set_tuning(8, sk)
            t = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), 30)
            out = psgd.precond_grad_kron(Ql, Qr, G)
            out2 = psgd.precond_grad_kron(Ql, Qr, G)
            res.append((t, ((out.double() - ref).norm() / ref.norm()).item(), torch.equal(out, out2)))
        print("%5dx%-5d apply %.3f -> %.3f ms (%+.0f%%)  rel err %.1e / %.1e  repeatable %s" %
              (M, N, res[0][0], res[1][0], (res[1][0] / res[0][0] - 1) * 100, res[0][1], res[1][1], res[1][2]))
    lib.psgd_kron_set_tuning(8, 1)
