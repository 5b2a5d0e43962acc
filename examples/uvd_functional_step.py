"""The UVd preconditioner through the FUNCTIONAL API on a least-squares toy problem, with the state placed by the library
(psgd_tf_amd/placement.py): the calls are the reference's (psgd.py:554, :619, fused as :732 -> :748), the state tensors U, V, d come
from `uvd_placed_state` instead of `tf.Variable`s, and the fused call writes its result into the arena's output region.

    python examples/uvd_functional_step.py [N] [r] [steps] [probe|packed]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import placement  # noqa: E402


def run(N=2_000_000, r=10, steps=200, mode="packed", lr=0.5, seed=0, device="cuda:0"):
    """minimise 0.5 * sum(c_i x_i^2) (a diagonal Hessian whose curvatures span a factor of 100) with preconditioned gradient steps
    x <- x - lr P g while P is being fitted (the preconditioner starts at 0.09 I, small enough for the stiffest direction, and learns
    the inverse curvatures: the loss falls by more than 1e4 x in 200 steps at N = 2M); returns the losses"""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    c = torch.exp(torch.empty(N, 1, device=dev).uniform_(-2.3, 2.3, generator=g))
    x = torch.randn(N, 1, device=dev, generator=g)
    U, V, d, arena = placement.uvd_placed_state(N, r, dev, preconditioner_init_scale=0.3, placement=mode)   # psgd.py:687-690, placed
    gen = torch.Generator().manual_seed(seed)                                       # the coins of :562, :588
    losses = []
    for _ in range(steps):
        losses.append(float(0.5 * torch.sum(c * x * x)))
        grad = c * x                                                                # gradient
        v = torch.randn(N, 1, device=dev, generator=g)                              # probe vector (:713)
        h = c * v                                                                   # Hessian-vector product
        pre = psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, grad, 0.1, psgd._tiny, generator=gen, out=arena.out)
        x = x - lr * pre
    losses.append(float(0.5 * torch.sum(c * x * x)))
    return losses, arena


if __name__ == "__main__":
    a = sys.argv[1:]
    losses, arena = run(int(a[0]) if a else 2_000_000, int(a[1]) if len(a) > 1 else 10, int(a[2]) if len(a) > 2 else 200,
                        a[3] if len(a) > 3 else "packed")
    print("layout: %s   loss %.4g -> %.4g in %d steps" % (arena.info.get("layout"), losses[0], losses[-1], len(losses) - 1))
