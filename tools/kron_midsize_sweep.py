"""Mid-size fp32 Kron (where the exact fp32-MFMA kernels run by default): default tile choice against forcing the 128-tile
split GEMM (psgd_kron_set_tuning(0, 2)) and against 64-tiles (0, 1).    python tools/kron_midsize_sweep.py"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for n in (256, 384, 512, 640, 768, 896, 1000, 1024, 1536, 2048):
        Ql, Qr = tri(n, g), tri(n, g)
        dX = torch.randn(n, n, device="cuda", generator=g)
        dG = dX * 1.5 + 0.1 * torch.randn(n, n, device="cuda", generator=g)
        G = torch.randn(n, n, device="cuda", generator=g)
        row = []
        for force in (0, 1, 2):
            lib.psgd_kron_set_tuning(0, force)
            def cold():
                Ql.add_(0.0)
                return psgd.precond_grad_kron(Ql, Qr, G)
            ta = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), 30)
            tc = timeit(cold, 30)
            tu = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 20)
            row.append("%s: apply %.3f (new factors %.3f) update %.3f" % (("auto", "64-tile", "128-tile/x3")[force], ta, tc, tu))
        print("%4d^2  " % n + " | ".join(row))
    lib.psgd_kron_set_tuning(0, 0)
