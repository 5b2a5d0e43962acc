"""Whole-call timing of the sparse-LU path on one GPU (development aid, not the bench contract).
Per-kernel numbers: run under `rocprofv3 --kernel-trace --stats` and read with tools/rocpd_stats.py."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=50_000_000)
    ap.add_argument("--r", type=int, default=10)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--nt", type=int, default=0, help="streaming policy (tuning key 0): 0 auto, 1 never nt, 2 always nt")
    ap.add_argument("--bpc", type=int, default=0, help="blocks per CU (tuning key 1)")
    args = ap.parse_args()
    N, r = args.N, args.r
    dev = torch.device("cuda:0")
    from psgd_tf_amd import _lib
    _lib.load().psgd_set_tuning(0, args.nt)
    _lib.load().psgd_set_tuning(1, args.bpc)
    g = torch.Generator(device=dev).manual_seed(0)
    n2 = N - r
    sc = 0.3 / r ** 0.5
    L12 = torch.randn(N, r, device=dev, generator=g) * (sc * 3 * (r / N) ** 0.5)
    U12 = torch.randn(r, N, device=dev, generator=g) * (sc * 3 * (r / N) ** 0.5)
    L12[:r] = torch.tril(torch.randn(r, r, device=dev, generator=g) * sc, -1) + torch.eye(r, device=dev)
    U12[:, :r] = torch.triu(torch.randn(r, r, device=dev, generator=g) * sc, 1) + torch.eye(r, device=dev)
    l3 = torch.exp(torch.empty(n2, 1, device=dev).uniform_(-0.5, 0.5, generator=g))
    u3 = torch.exp(torch.empty(n2, 1, device=dev).uniform_(-0.5, 0.5, generator=g)) * 0.7
    dx = torch.randn(N, 1, device=dev, generator=g)
    dg = dx * torch.exp(torch.empty(N, 1, device=dev).uniform_(-2.3, 2.3, generator=g))
    gr = torch.randn(N, 1, device=dev, generator=g)
    calls = {
        "apply": (lambda: psgd.precond_grad_splu(L12, l3, U12, u3, [gr]), 4 * (3 * r + 9)),
        "update": (lambda: psgd.update_precond_splu(L12, l3, U12, u3, [dx], [dg], 0.01), 4 * (9 * r + 15)),
    }
    for name, (fn, bpr) in calls.items():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        gbs = bpr * N / (ms * 1e-3) / 1e9
        print("%-7s N=%d r=%d %9.3f ms  %8.1f GB/s of actual traffic (%.1f%% of 8 TB/s)  %.2f Gparam/s" %
              (name, N, r, ms, gbs, gbs / 80.0, N / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
