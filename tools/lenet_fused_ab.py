"""LeNet5 layer set (mnist_with_lenet5.py:12-16) through the reference's per-layer calls: fused strip kernels (tuning key 21 = 1)
against the stage kernels (0), eager and as a captured graph of the two list comprehensions of mnist_with_lenet5.py:51,53, next to
the batched extension.  Prints us per set; per-layer us with --layers."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import _lib, kron

LENET5 = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10)]
dev = torch.device("cuda:0")
lib = _lib.load()


def state(M, N):
    g = torch.Generator(device=dev).manual_seed(M * 7 + N)
    Ql = torch.triu(torch.randn(M, M, device=dev, generator=g) * 0.02, 1) + torch.eye(M, device=dev)
    Qr = torch.triu(torch.randn(N, N, device=dev, generator=g) * 0.02, 1) + torch.eye(N, device=dev)
    return Ql, Qr, torch.randn(M, N, device=dev, generator=g), torch.randn(M, N, device=dev, generator=g)


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def graphed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    return g.replay


sts = [state(m, n) for m, n in LENET5]
shapes = LENET5 + [(int(a), int(b)) for a, b in (x.split("x") for x in sys.argv[1:] if "x" in x)]
for fused in (1, 0):
    kron.set_tuning(21, fused)
    ap = lambda: [psgd.precond_grad_kron(a, b, g) for a, b, g, x in sts]
    up = lambda: [psgd.update_precond_kron(a, b, x, g, 0.01) for a, b, g, x in sts]
    print("fused=%d  per-layer apply %.1f us (graph %.1f)   update %.1f us (graph %.1f)"
          % (fused, timeit(ap), timeit(graphed(ap)), timeit(up), timeit(graphed(up))))
    if "--layers" in sys.argv:
        for (m, n) in shapes:
            a, b, g, x = state(m, n)
            fa = lambda: psgd.precond_grad_kron(a, b, g)
            fu = lambda: psgd.update_precond_kron(a, b, x, g, 0.01)
            print("   %4d x %-4d apply %.1f us (graph %.1f)   update %.1f us (graph %.1f)"
                  % (m, n, timeit(fa), timeit(graphed(fa)), timeit(fu), timeit(graphed(fu))))
kron.set_tuning(21, 0)
Qls, Qrs, Gs, Xs = ([s[i] for s in sts] for i in range(4))
print("batched apply %.1f us   batched update %.1f us" % (timeit(lambda: psgd.precond_grad_kron_batched(Qls, Qrs, Gs)),
                                                        timeit(lambda: psgd.update_precond_kron_batched(Qls, Qrs, Xs, Gs, 0.01))))
