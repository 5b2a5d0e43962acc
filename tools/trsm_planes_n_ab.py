"""From which factor size on the solves of the fp32 Kron update should use the factors' planes (tuning key 15).
    python tools/trsm_planes_n_ab.py
"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for M, N in ((1024, 1024), (1536, 1536), (2048, 2048), (1300, 1300), (2048, 1024), (1000, 3000), (1024, 4096)):
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5
        res = {}
        for rnd in range(3):
            for mn in (2048, 512):
                lib.psgd_kron_set_tuning(15, mn)
                t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 10)
                res[mn] = min(res.get(mn, 1e9), t)
        print("%dx%d: solves on planes only above 2048: %.3f ms   above 512: %.3f ms" % (M, N, res[2048], res[512]))
    lib.psgd_kron_set_tuning(15, 2048)
