"""A/B of the strip grouping of the blocked triangular solves (tuning key 5) inside the fp32 and the bf16-operand update.
    python tools/trsm_group_ab.py [M N]"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    shapes = [(4096, 4096), (2048, 2048), (2048, 8192)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for M, N in shapes:
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
        dXb, dGb = dX.bfloat16(), dG.bfloat16()
        ref = None
        for grp in (1, 2, 4, 1, 2, 4):
            lib.psgd_kron_set_tuning(5, grp)
            t32 = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 10)
            tb = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01), 10)
            out = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
            if ref is None:
                ref = out
            d = max(((a - b).norm() / b.norm()).item() for a, b in zip(out, ref))
            print("Kron update %dx%d strips/group=%d: fp32 %.3f ms, bf16 operands %.3f ms, rel diff of factors vs group=1 %.1e" % (M, N, grp, t32, tb, d))
    lib.psgd_kron_set_tuning(5, 2)
