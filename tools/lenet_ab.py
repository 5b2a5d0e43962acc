"""A/B of the 32-tile small-GEMM bodies on the batched LeNet5 set (run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_timing import state  # noqa: E402

dev = torch.device("cuda:0")
LENET5 = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10)]
sts = [state(m, n, dev) for m, n in LENET5]
Qls, Qrs, dXs, dGs, Gs = ([s[i] for s in sts] for i in range(5))
lib = _lib.load()


def timeit(f, n=200):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out


ref = None
for name, k in (("gemm_body<32,64>", 0), ("k_gemm_small (ring of 4 K tiles, precomputed offsets)", 1)):
    lib.psgd_kron_set_tuning(3, k)
    ta, oa = timeit(lambda: psgd.precond_grad_kron_batched(Qls, Qrs, Gs))
    tu, ou = timeit(lambda: psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01))
    if ref is None:
        ref = (oa, ou)
    d = max(max(((a - b).norm() / b.norm()).item() for a, b in zip(oa, ref[0])),
            max(((a - b).norm() / b.norm()).item() for x, y in zip(ou, ref[1]) for a, b in zip(x, y)))
    print("%-56s apply %.1f us  update %.1f us  (max rel diff vs first %.1e)" % (name, ta, tu, d))
lib.psgd_kron_set_tuning(3, 1)
