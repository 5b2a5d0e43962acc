"""A/B of the three-term split in the trailing products of the bf16-operand Kron update's solves (tuning key 3).
    python tools/trsm_lite_ab.py [M N]"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    Ql, Qr = tri(M, g), tri(N, g)
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
    dXb, dGb = dX.bfloat16(), dG.bfloat16()
    a = psgd.update_precond_kron(Ql, Qr, dXb.float(), dGb.float(), 0.01)
    rho = (Ql.diagonal().max() / Qr.diagonal().max()).sqrt()
    base = (Ql / rho, Qr * rho)
    for rep in range(2):
        for lite in (0, 1):
            lib.psgd_kron_bf16_set_tuning(3, lite)
            t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01), 20)
            b = psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01)
            err = max(((x - y).norm() / y.norm()).item() for x, y in zip(b, a))
            inc = max((((x - z) - (y - z)).norm() / (y - z).norm()).item() for x, y, z in zip(b, a, base))
            print("Kron bf16 update %dx%d lite=%d: %.3f ms, state rel diff vs fp32 path %.2e, increment rel diff %.2e" % (M, N, lite, t, err, inc))
