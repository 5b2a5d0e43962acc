"""A/B of the triangular-solve strip kernels inside a 4096^2 Kron update (run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_timing import state  # noqa: E402

M = N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Ql, Qr, dX, dG, G = state(M, N, torch.device("cuda:0"))
lib = _lib.load()


def timeit(n=10):
    psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


ref = None
for name, k2 in (("register-resident strips", 0), ("LDS-resident strips", 1)):
    lib.psgd_kron_set_tuning(2, k2)
    t, out = timeit()
    if ref is None:
        ref = out
    d = max(((a - b).norm() / b.norm()).item() for a, b in zip(out, ref))
    print("%-42s update %dx%d fp32: %.3f ms   (max rel diff vs first %.1e)" % (name, M, N, t, d))
lib.psgd_kron_set_tuning(2, 0)
