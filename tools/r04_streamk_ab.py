"""Round 4: the bf16-operand Kron update with its products as stream-K launches (psgd_kron_bf16_set_tuning key 4: 1, default) against the
one-tile-per-workgroup 128^2 kernels (0).   python tools/r04_streamk_ab.py"""
import sys
import torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402
from tools.kron_f16_planes_ab import update_ref64, errs  # noqa: E402
lib = _lib.load()
g = torch.Generator(device="cuda"); g.manual_seed(1)
for M, N in ((4096, 4096), (2048, 4096), (2048, 2048), (3072, 3072), (6144, 6144), (8192, 2048)):
    Ql, Qr = tri(M, g), tri(N, g)
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = dX * torch.exp(torch.rand(M, 1, device="cuda", generator=g) * 2 - 1) * torch.exp(torch.rand(1, N, device="cuda", generator=g) * 2 - 1)
    dXb, dGb = dX.bfloat16(), dG.bfloat16()
    rl, rr, bl, br = update_ref64(Ql, Qr, dXb.float(), dGb.float(), 0.01)
    res = {}
    for rnd in range(2):
        for key in (0, 1):
            lib.psgd_kron_bf16_set_tuning(4, key)
            t = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01), 8) for _ in range(2))
            if rnd == 0:
                a, b = psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01)
                res[key] = [t, errs(a, rl)[0], errs(b, rr)[0], errs(a, rl)[1] if len(errs(a, rl)) > 1 else 0]
            else:
                res[key][0] = min(res[key][0], t)
    lib.psgd_kron_bf16_set_tuning(4, 1)
    print("%-10s bf16-operand update  128^2 tiles: %.3f ms   stream-K 256^2: %.3f ms   state rel err %.1e/%.1e -> %.1e/%.1e"
          % ("%dx%d" % (M, N), res[0][0], res[1][0], res[0][1], res[0][2], res[1][1], res[1][2]), flush=True)
