"""Round 6: the fp32 Kron calls over layer-like shapes (update, apply with new factors on the default route), with the issued-flop rate of
each -- where the time of the mid sizes goes.   python tools/r06_kron_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from r06_kron_ab import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for M, N in ((512, 512), (768, 768), (1024, 1024), (1536, 1536), (2048, 2048), (512, 2048), (1024, 4096), (768, 3072), (2048, 8192), (4096, 1024)):
        g = torch.Generator(device=dev).manual_seed(M * 7 + N)
        mk = lambda n: torch.triu(torch.randn(n, n, device=dev, generator=g) * 0.02, 1) + torch.eye(n, device=dev)
        pairs = [(mk(M), mk(N)), (mk(M), mk(N))]
        G, dX = torch.randn(M, N, device=dev, generator=g), torch.randn(M, N, device=dev, generator=g)
        flip = [0]

        def cold():
            flip[0] ^= 1
            return psgd.precond_grad_kron(pairs[flip[0]][0], pairs[flip[0]][1], G)
        tu = min(timeit(lambda: psgd.update_precond_kron(pairs[0][0], pairs[0][1], dX, G, 0.01), 10, warm_ms=60.0) for _ in range(3))
        ta = min(timeit(cold, 10, warm_ms=60.0) for _ in range(3))
        # triangular-aware flops: update = A (M^2 N + M N^2) + solves (same) + gradients (M^2 N + M N^2) x 2 ... = 4 (M^2 N + M N^2) + (M^3 + N^3) / 3 * 2 (factor updates), MACs x 2
        fu = 2.0 * (4.0 * (M * M * N + M * N * N) / 1.0 * 0.5 * 2 + (M**3 + N**3) / 3.0)
        fa = 2.0 * (M * M * N + M * N * N)            # four triangular products
        print("%5d x %5d  update %.3f ms (%.0f TFLOP/s on the triangular count)  apply (new factors) %.3f ms (%.0f TFLOP/s)"
              % (M, N, tu, fu / tu / 1e9, ta, fa / ta / 1e9), flush=True)


if __name__ == "__main__":
    main()
