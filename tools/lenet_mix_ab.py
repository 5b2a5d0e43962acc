"""A/B of the mixed product+solve stages (kron tuning key 7) on the batched LeNet5-set update (run on the GPU box)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import _lib
from tools.kron_timing import state

dev = torch.device("cuda:0")
LENET5 = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10)]
sts = [state(m, n, dev) for m, n in LENET5]
Qls, Qrs, dXs, dGs, Gs = ([s[i] for s in sts] for i in range(5))


def run(n=300):
    for _ in range(20):
        psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


lib = _lib.load()
outs = {}
for key in (0, 1, 0, 1):
    lib.psgd_kron_set_tuning(7, key)
    print("mix", key, "batched update %.1f us" % run(), flush=True)
    outs[key] = [(a.clone(), b.clone()) for a, b in psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01)]
same = all(torch.equal(a, b) for pa, pb in zip(outs[0], outs[1]) for a, b in zip(pa, pb))
print("bit-identical results:", same)
