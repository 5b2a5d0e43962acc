"""Timing of the Kron dense(x)dense path: LeNet5 layer set (mnist_with_lenet5.py:12-16) and 4096x4096."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402

LENET = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10)]


def flops_apply(M, N):
    return 2 * M**3 + 2 * M * M * N + 4 * M * N * N if M < N else 2 * N**3 + 2 * M * N * N + 4 * M * M * N


def flops_update(M, N):
    return 7 * (M * M * N + M * N * N) + 2 * (M**3 + N**3)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def state(M, N, dev):
    g = torch.Generator(device=dev).manual_seed(M * 7 + N)
    Ql = torch.triu(torch.randn(M, M, device=dev, generator=g) * 0.02, 1) + torch.eye(M, device=dev)
    Qr = torch.triu(torch.randn(N, N, device=dev, generator=g) * 0.02, 1) + torch.eye(N, device=dev)
    dX = torch.randn(M, N, device=dev, generator=g)
    dG = torch.randn(M, N, device=dev, generator=g)
    G = torch.randn(M, N, device=dev, generator=g)
    return Ql, Qr, dX, dG, G


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sts = [state(M, N, dev) for M, N in LENET]

    def lenet_apply():
        return [psgd.precond_grad_kron(Ql, Qr, G) for (Ql, Qr, dX, dG, G) in sts]

    def lenet_update():
        return [psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01) for (Ql, Qr, dX, dG, G) in sts]

    Qls, Qrs = [x[0] for x in sts], [x[1] for x in sts]
    dXs, dGs, Gs = [x[2] for x in sts], [x[3] for x in sts], [x[4] for x in sts]
    tab = timeit(lambda: psgd.precond_grad_kron_batched(Qls, Qrs, Gs), 50)
    tub = timeit(lambda: psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01), 50)
    fa = sum(flops_apply(M, N) for M, N in LENET)
    fu = sum(flops_update(M, N) for M, N in LENET)
    ta, tu = timeit(lenet_apply, 50), timeit(lenet_update, 50)
    print("LeNet5 set  apply  %8.1f us  %7.2f GFLOP/s (F_ref)" % (ta * 1e3, fa / ta / 1e6))
    print("LeNet5 set  update %8.1f us  %7.2f GFLOP/s (F_ref)" % (tu * 1e3, fu / tu / 1e6))
    print("LeNet5 set  apply  batched %8.1f us  %7.2f GFLOP/s" % (tab * 1e3, fa / tab / 1e6))
    print("LeNet5 set  update batched %8.1f us  %7.2f GFLOP/s" % (tub * 1e3, fu / tub / 1e6))
    for (M, N) in [(257, 120), (1024, 1024), (args.big, args.big)]:
        Ql, Qr, dX, dG, G = state(M, N, dev)
        ta = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), args.iters)
        tu = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), max(2, args.iters // 4))
        print("%5dx%-5d apply  %9.3f ms  %8.1f GFLOP/s   update %9.3f ms  %8.1f GFLOP/s (F_ref)" %
              (M, N, ta, flops_apply(M, N) / ta / 1e6, tu, flops_update(M, N) / tu / 1e6))
        if M % 8 == 0 and N % 8 == 0:
            Gb = G.to(torch.bfloat16)
            tb = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, Gb), args.iters)
            print("%5dx%-5d apply bf16 operands %9.3f ms  %8.1f GFLOP/s (F_ref)" % (M, N, tb, flops_apply(M, N) / tb / 1e6))


if __name__ == "__main__":
    main()
