"""Round 4: CU-masked streams for the large fp32 Kron update (tuning key 26 = CUs of each of the two inversion-chain streams; the products
of :173 run on the other 256 - 2c CUs).  One masked width per process (the streams are created once): pass the widths as arguments,
the script runs itself once per width.
    python tools/r04_cumask_ab.py            # parent: widths 0 16 32 48 64
    python tools/r04_cumask_ab.py one <c>    # child"""
import subprocess
import sys
if len(sys.argv) < 2 or sys.argv[1] != "one":
    for c in (sys.argv[1:] or ["0", "16", "32", "48", "64"]):
        subprocess.run([sys.executable, __file__, "one", c], check=False)
    sys.exit(0)
import torch  # noqa: E402
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402
from tools.kron_f16_planes_ab import update_ref64, errs  # noqa: E402
c = int(sys.argv[2])
lib = _lib.load()
lib.psgd_kron_set_tuning(26, c)
g = torch.Generator(device="cuda"); g.manual_seed(1)
for M, N, order in ((4096, 4096, -1), (2048, 4096, 1), (3072, 3072, 1), (6144, 6144, 1), (2048, 2048, 1)):
    lib.psgd_kron_set_tuning(25, order)
    Ql, Qr = tri(M, g), tri(N, g)
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = dX * torch.exp(torch.rand(M, 1, device="cuda", generator=g) * 2 - 1) * torch.exp(torch.rand(1, N, device="cuda", generator=g) * 2 - 1)
    rl, rr, bl, br = update_ref64(Ql, Qr, dX, dG, 0.01)
    a, b = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
    t = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 8) for _ in range(3))
    for tag, dt in (("bf16", torch.bfloat16),):
        tb = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX.to(dt), dG.to(dt), 0.01), 8) for _ in range(3)) if M * N <= 4096 * 4096 else float("nan")
    print("c=%-3d %-10s order %2d  fp32 %.3f ms  rel %.1e/%.1e   bf16 operands %.3f ms" % (c, "%dx%d" % (M, N), order, t, errs(a, rl)[0], errs(b, rr)[0], tb), flush=True)
