"""Repeatability soak of the K-split hand-off (write-through partials + agent-scope ticket): the same call many times, every result
compared bit for bit with the first.   python tools/splitk_soak.py [iterations]"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from tools.kron_bf16_update_timing import tri  # noqa: E402

if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    bad = 0
    for M, N, what in ((64, 8192, "apply"), (130, 5000, "apply"), (200, 3072, "apply"), (4096, 4096, "update"), (2944, 2944, "update"),
                       (600, 5200, "update")):
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
        f = (lambda: (psgd.precond_grad_kron(Ql, Qr, dX),)) if what == "apply" else (lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01))
        first = [t.clone() for t in f()]
        n = iters if what == "apply" else max(iters // 6, 20)
        mism = 0
        for _ in range(n):
            out = f()
            if not all(torch.equal(a, b) for a, b in zip(out, first)):
                mism += 1
        bad += mism
        print("%s %dx%d: %d calls, %d differ from the first" % (what, M, N, n, mism))
    sys.exit(1 if bad else 0)
