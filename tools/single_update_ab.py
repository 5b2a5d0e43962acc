"""Single small-layer Kron update: the large-layer path against the batch-of-one route (kron tuning key 7, bit 1)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import preconditioned_stochastic_gradient_descent as psgd
from tools.kron_timing import state
from psgd_tf_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
SH = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10), (400, 300), (512, 512), (64, 500), (512, 40), (300, 300), (2, 3)]
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for m, n in SH:
    Ql, Qr, dX, dG, G = state(m, n, dev)
    lib.psgd_kron_set_tuning(7, 1)
    a = t(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01))
    r1 = [x.clone() for x in psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)]
    lib.psgd_kron_set_tuning(7, 3)
    b = t(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01))
    r2 = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
    print(f"{m}x{n}: large-layer path {a:.1f} us, batch of one {b:.1f} us, equal {torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1])}, maxdiff {max((r1[0]-r2[0]).abs().max().item(), (r1[1]-r2[1]).abs().max().item()):.2e}", flush=True)
