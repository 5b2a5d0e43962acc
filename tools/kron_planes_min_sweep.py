"""fp32 Kron around the planes thresholds: the shape rules of kron_planes / kron_planes_apply against the first rule of the
round (M, N >= 1024; PSGD_KRON_PLANES_OLD=1, read once per process).   [PSGD_KRON_PLANES_OLD=1] python tools/kron_planes_min_sweep.py"""
import os
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    shapes = ((512, 512), (640, 640), (896, 896), (1000, 1000), (900, 1400), (520, 3000), (300, 4000), (256, 8192), (384, 2048),
              (200, 6000), (1024, 1024))
    if os.environ.get("MID"):
        shapes = ((384, 700), (500, 500), (500, 700), (700, 700), (384, 1000), (500, 1000), (700, 1000), (384, 1300), (500, 1300),
                  (384, 1700), (500, 1700), (200, 1700), (384, 2500), (200, 2500), (200, 3072), (256, 1024), (300, 300))
    if os.environ.get("SPLITK2"):
        shapes = ((500, 1700), (1000, 1000), (384, 1300), (700, 1000), (500, 1300), (384, 1700), (1024, 1024), (200, 1700), (900, 1400),
                  (640, 1280), (1300, 500))
    if os.environ.get("SPLITK"):
        shapes = ((200, 3072), (384, 2500), (200, 2500), (256, 2048), (384, 2048), (130, 3000), (100, 2100), (500, 1700), (64, 2048),
                  (128, 4096), (3072, 200), (2048, 256))
    if os.environ.get("ALIGNED"):
        shapes = ((128, 4096), (256, 2048), (384, 2048), (512, 1024), (256, 4096), (640, 640), (768, 768), (896, 896), (512, 2048),
                  (1024, 512), (640, 1280), (128, 8192))
    if os.environ.get("SKINNY"):
        shapes = ((200, 6000), (6000, 200), (64, 8192), (10, 8192), (128, 4096), (130, 5000), (30, 3000))
    for M, N in shapes:
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
        G = torch.randn(M, N, device="cuda", generator=g)
        def cold():
            Ql.add_(0.0)
            return psgd.precond_grad_kron(Ql, Qr, G)
        ta = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), 30)
        tc = timeit(cold, 30)
        tu = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 20)
        print("%s  %4dx%-4d apply %.3f (new factors %.3f) update %.3f ms" % (("old rule" if os.environ.get("PSGD_KRON_PLANES_OLD") == "1" else "t128>=%s upd>=%s updt>=%s" % (os.environ.get("PSGD_KRON_PLANES_T128", "64"), os.environ.get("PSGD_KRON_PLANES_UPD", "512"), os.environ.get("PSGD_KRON_PLANES_UPD_T128", "64"))), M, N, ta, tc, tu))
