"""A/B of the large fp32 Kron update: GEMM stages on operand planes (tuning key 4 = 1) against the in-GEMM split.
    python tools/kron_planes_update_ab.py [M N]"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    shapes = [(4096, 4096), (2048, 2048), (1024, 1024), (2048, 8192)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for M, N in shapes:
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
        for rep in range(2):
            for planes in (0, 1):
                lib.psgd_kron_set_tuning(4, planes)
                t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 10)
                print("Kron fp32 update %dx%d planes=%d: %.3f ms" % (M, N, planes, t))
    lib.psgd_kron_set_tuning(4, 1)
