"""Which update products of the blocked triangular solves (fp32 Kron update, f16 x 2 planes) should run on planes:
tuning keys 5 (strips per group), 13 (least K on planes), 14 (least output tiles on planes).

    python tools/trsm_planes_k_ab.py
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
    settings = [(0, 2048, 256), (0, 512, 256), (0, 512, 64), (2, 1024, 256), (2, 512, 256), (2, 512, 64), (8, 512, 256), (8, 512, 64),
                (1, 512, 64)]
    for M, N in ((4096, 4096), (2560, 2560), (2048, 4096), (8192, 1024), (3072, 3072), (6144, 6144)):
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5
        base = None
        out = []
        for rnd in range(2):
            for i, (grp, mink, mint) in enumerate(settings):
                lib.psgd_kron_set_tuning(5, grp); lib.psgd_kron_set_tuning(13, mink); lib.psgd_kron_set_tuning(14, mint)
                t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 8)
                a, b = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
                if base is None:
                    base = (a, b)
                err = max(((a - base[0]).norm() / base[0].norm()).item(), ((b - base[1]).norm() / base[1].norm()).item())
                if rnd == 0:
                    out.append([t, err])
                else:
                    out[i][0] = min(out[i][0], t)
        print("%dx%d: " % (M, N) + "  ".join("g%d/k%d/t%d %.3f ms (%.0e)" % (s + (o[0], o[1])) for s, o in zip(settings, out)))
    lib.psgd_kron_set_tuning(5, 0); lib.psgd_kron_set_tuning(13, 2048); lib.psgd_kron_set_tuning(14, 256)
