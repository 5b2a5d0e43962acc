"""Round 4: tile order of the factor-update launch (psgd.py:179): whole tile rows per XCD against 4 x 4 tile patches (fp32: tuning key 27, bf16 operands: bf16 key 5)."""
import sys, torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import _lib
from tools.kron_bf16_update_timing import tri, timeit
lib = _lib.load()
g = torch.Generator(device="cuda"); g.manual_seed(1)
for M, N in ((4096, 4096), (6144, 6144), (4096, 8192), (5120, 4096)):
    Ql, Qr = tri(M, g), tri(N, g)
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = dX * 1.5
    res = {}
    for rnd in range(2):
        for key in (0, 1):
            lib.psgd_kron_set_tuning(27, key)
            t = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 8) for _ in range(2))
            if rnd == 0:
                res[key] = [t, psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)]
            else:
                res[key][0] = min(res[key][0], t)
    lib.psgd_kron_set_tuning(27, 0)
    same = all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
    print("%dx%d fp32 update: row order %.3f ms, patch order %.3f ms, bitwise equal %s" % (M, N, res[0][0], res[1][0], same), flush=True)
    if M * N <= 6144 * 6144:
        dXb, dGb = dX.bfloat16(), dG.bfloat16()
        res = {}
        for rnd in range(2):
            for key in (0, 1):
                lib.psgd_kron_bf16_set_tuning(5, key)
                t = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01), 8) for _ in range(2))
                if rnd == 0:
                    res[key] = [t, psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01)]
                else:
                    res[key][0] = min(res[key][0], t)
        lib.psgd_kron_bf16_set_tuning(5, 0)
        same = all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
        print("%dx%d bf16-operand update: row order %.3f ms, patch order %.3f ms, bitwise equal %s" % (M, N, res[0][0], res[1][0], same), flush=True)
