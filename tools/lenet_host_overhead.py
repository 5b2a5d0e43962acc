"""Host-side enqueue time vs. device time of the small-layer Kron calls (development aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402

LENET5 = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10)]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
tri = lambda n: torch.triu(torch.randn(n, n, device=dev, generator=g) * 0.02, 1) + torch.eye(n, device=dev)
sts = [(tri(m), tri(n), torch.randn(m, n, device=dev, generator=g)) for m, n in LENET5]
Qls, Qrs, Gs = [x[0] for x in sts], [x[1] for x in sts], [x[2] for x in sts]
dXs = [torch.randn_like(x) for x in Gs]


def measure(name, fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-34s host enqueue %6.1f us   total %6.1f us per call" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))


measure("per-layer apply x5 (cached Grams)", lambda: [psgd.precond_grad_kron(a, b, c) for a, b, c in sts])
measure("batched apply (cached Grams)", lambda: psgd.precond_grad_kron_batched(Qls, Qrs, Gs))
measure("batched update", lambda: psgd.update_precond_kron_batched(Qls, Qrs, dXs, Gs, 0.01))
measure("per-layer update x5", lambda: [psgd.update_precond_kron(a, b, x, c, 0.01) for (a, b, c), x in zip(sts, dXs)])
