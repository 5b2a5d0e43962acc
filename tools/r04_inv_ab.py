"""Round 4, the inversions of the large fp32 / bf16-operand Kron update: 512-blocks from ONE strip launch (tuning key 23) and the K
blocked solves on the inverses of diagonal h-blocks (key 24: h; 0 = whole inverses, one product per solve), against their predecessors; time and errors of the new factors / increments
against an fp64 run.      python tools/r04_inv_ab.py [quick]"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402
from tools.kron_f16_planes_ab import update_ref64, errs  # noqa: E402

if __name__ == "__main__":
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    shapes = ((4096, 4096), (2048, 2048), (3072, 3072), (2048, 4096), (2944, 2944), (6144, 6144))
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        shapes = ((4096, 4096), (2048, 4096))
    combos = ((0, 0), (1, 0), (1, 2048), (1, 1024))              # (key 23: 512-blocks from one strip launch, key 24: block size of the blocked solves; 0 = whole inverses)
    for bf16 in (False, True):
        for M, N in shapes:
            Ql, Qr = tri(M, g), tri(N, g)
            dX = torch.randn(M, N, device="cuda", generator=g)
            dG = dX * torch.exp(torch.rand(M, 1, device="cuda", generator=g) * 2 - 1) * torch.exp(torch.rand(1, N, device="cuda", generator=g) * 2 - 1)
            if bf16:
                dX, dG = dX.bfloat16(), dG.bfloat16()
            rl, rr, bl, br = update_ref64(Ql, Qr, dX.float(), dG.float(), 0.01)
            res = {}
            for rnd in range(2):
                for c in combos:
                    lib.psgd_kron_set_tuning(23, c[0]); lib.psgd_kron_set_tuning(24, c[1])
                    t = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 8)
                    if rnd == 0:
                        a, b = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
                        a2, b2 = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
                        il = ((a.double() - bl) - (rl - bl)).norm() / (rl - bl).norm()
                        ir = ((b.double() - br) - (rr - br)).norm() / (rr - br).norm()
                        res[c] = [t, errs(a, rl)[0], errs(b, rr)[0], il.item(), ir.item(), torch.equal(a, a2) and torch.equal(b, b2)]
                    else:
                        res[c][0] = min(res[c][0], t)
            print("%-10s %s update  " % ("%dx%d" % (M, N), "bf16-operand" if bf16 else "fp32") +
                  "  ".join("strip512=%d blk=%d: %.3f ms" % (c[0], c[1], res[c][0]) for c in combos))
            print("           rel %s   increment %s   rep %s" % (
                " ".join("%.1e/%.1e" % (res[c][1], res[c][2]) for c in combos),
                " ".join("%.1e/%.1e" % (res[c][3], res[c][4]) for c in combos), [res[c][5] for c in combos]))
    lib.psgd_kron_set_tuning(23, 1); lib.psgd_kron_set_tuning(24, 2048)
