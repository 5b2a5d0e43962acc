"""Increment error of the fp32 Kron update near its fixed point with ill-conditioned factors (the case of
tests/test_kron_gpu.py::test_update_with_ill_conditioned_factors), per plane format (tuning key 12) and seed.
    python tools/illcond_increment_probe.py
"""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402


def ref_update(Ql, Qr, dX, dG, step):
    rho = np.sqrt(np.max(np.diag(Ql)) / np.max(np.diag(Qr)))
    Ql, Qr = Ql / rho, Qr * rho
    A = Ql @ (dG @ Qr.T)
    Bt = np.linalg.solve(Ql.T, np.linalg.solve(Qr.T, dX.T).T)
    g1, g2 = np.triu(A @ A.T - Bt @ Bt.T), np.triu(A.T @ A - Bt.T @ Bt)
    tiny = np.finfo(np.float32).tiny
    return (Ql - step / (np.abs(g1).max() + tiny) * g1 @ Ql, Qr - step / (np.abs(g2).max() + tiny) * g2 @ Qr), (Ql, Qr)


if __name__ == "__main__":
    lib = _lib.load()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    for M, N in ((1100, 530), (2048, 1536), (700, 600)):
        for seed in range(4):
            rng = np.random.default_rng(M + 13 * N + 1000 * seed)

            def illcond(n):
                d = np.exp(np.linspace(0.0, -np.log(1e4), n))
                rng.shuffle(d)
                return np.triu(rng.standard_normal((n, n)) * (0.3 / n ** 0.5), 1) * d[None, :] + np.diag(d)
            Ql, Qr = illcond(M), illcond(N)
            dX = rng.standard_normal((M, N))
            dG = np.linalg.solve(Ql.T @ Ql, dX) @ np.linalg.inv(Qr.T @ Qr) * np.exp(rng.uniform(-0.5, 0.5, (1, N)))
            a32 = [x.astype(np.float32) for x in (Ql, Qr, dX, dG)]
            ref, base = ref_update(*(x.astype(np.float64) for x in a32), 0.01)
            out = []
            for key in (0, 2):
                lib.psgd_kron_set_tuning(12, key)
                got = psgd.update_precond_kron(*(dev(x) for x in a32), 0.01)
                e = [np.linalg.norm((g.cpu().numpy().astype(np.float64) - b) - (r - b)) / np.linalg.norm(r - b) for g, r, b in zip(got, ref, base)]
                out.append("key12=%d: %.1e / %.1e" % (key, e[0], e[1]))
            print("%dx%d seed %d  increment error  %s" % (M, N, seed, "   ".join(out)))
    lib.psgd_kron_set_tuning(12, 2)
