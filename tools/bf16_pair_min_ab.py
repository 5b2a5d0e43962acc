import os, sys, torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd
from tools.kron_bf16_update_timing import tri, timeit
g = torch.Generator(device="cuda"); g.manual_seed(0)
shapes = ((1536, 1536), (2048, 2048), (2560, 2560), (2500, 2500), (2048, 3072), (3072, 3072), (2304, 2304), (1792, 2816))
if os.environ.get("ODD"):
    shapes = ((1700, 1300), (3072, 384), (3072, 1300), (1300, 3072), (1300, 1700), (384, 3072))
if os.environ.get("SMALL"):
    shapes = ((512, 512), (768, 768), (1024, 1024), (1280, 1280), (1024, 2048), (512, 2048), (256, 4096), (1280, 1536), (768, 1536), (1024, 1536), (256, 1024))
for M, N in shapes:
    Ql, Qr = tri(M, g), tri(N, g)
    G = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    ref = (Ql.double().T @ Ql.double()) @ G.double() @ (Qr.double().T @ Qr.double())
    t = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), 40)
    out = psgd.precond_grad_kron(Ql, Qr, G)
    print("min_tiles=%s  %dx%d bf16 apply %.3f ms  rel err %.2e" % (os.environ.get("PSGD_BF16_PAIR_MIN_TILES", "128"), M, N, t, ((out.double() - ref).norm() / ref.norm()).item()))
