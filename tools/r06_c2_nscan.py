"""Config 2 family (r = 10): per-kernel time of the fused step against N -- what part of each sweep is a fixed cost?"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402

TINY = 1.1754943508222875e-38


def main():
    r = int(os.environ.get("C2_R", 10))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for N in (62_500, 125_000, 250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000):
        g = torch.Generator(device=dev).manual_seed(7)
        sc = (1.0 / (N * r)) ** 0.5
        U, V = torch.randn(N, r, device=dev, generator=g) * sc, torch.randn(N, r, device=dev, generator=g) * sc
        d = torch.ones(N, 1, device=dev)
        gr, v = torch.randn(N, 1, device=dev, generator=g), torch.randn(N, 1, device=dev, generator=g)
        h = v * 1.5
        out = torch.empty_like(gr)

        def step(i):
            return psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, gr, 0.0, TINY, balance=False, update_U=(i % 2 == 0), out=out)
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        lib.psgd_prof_enable(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            step(i)
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) / 200 * 1e3
        lib.psgd_prof_enable(1)
        for i in range(100):
            step(i)
        torch.cuda.synchronize()
        ks = []
        for slot in (3, 4, 2):
            tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
            lib.psgd_prof_collect(slot, ctypes.byref(tot), ctypes.byref(cnt))
            ks.append(tot.value / max(cnt.value, 1) * 1e3)
        lib.psgd_prof_enable(0)
        print("N %8d : step %7.1f us | gram %6.1f  s2 %6.1f  final %6.1f  | the rest (serial kernels, gaps) %5.1f" %
              (N, wall, ks[0], ks[1], ks[2], wall - sum(ks)), flush=True)
        del U, V, d, gr, v, h, out


if __name__ == "__main__":
    main()
