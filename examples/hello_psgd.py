"""Hello-world of PSGD on the Rosenbrock function: this repository's counterpart of the
reference harness hello_psgd.py:7-27 (torch.autograd in place of tf.GradientTape; the
reference script itself cannot travel).  Dense 2x2 preconditioner, eager, CPU -- config 1 of
BASELINE.json (plumbing, no GPU).

    xs = (-1, 1)  (:7)   Q = 0.1 I  (:8)   500 iterations  (:15)   step = 0.2  (:25)   lr = 0.5  (:27)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402


def rosenbrock(xs):                                                       # hello_psgd.py:10-12
    x1, x2 = xs
    return 100.0 * (x2 - x1 ** 2) ** 2 + (1.0 - x1) ** 2


def run(num_iter=500, seed=0, dtype=torch.float32, first_v=None):
    gen = torch.Generator().manual_seed(seed)
    xs = [torch.tensor(-1.0, dtype=dtype, requires_grad=True), torch.tensor(1.0, dtype=dtype, requires_grad=True)]
    Q = 0.1 * torch.eye(2, dtype=dtype)
    f_values = []
    for it in range(num_iter):
        y = rosenbrock(xs)
        grads = torch.autograd.grad(y, xs, create_graph=True)            # 1st derivatives (:16-19)
        if it == 0 and first_v is not None:
            vs = [torch.tensor(v, dtype=dtype) for v in first_v]
        else:
            vs = [torch.randn(x.shape, generator=gen, dtype=dtype) for x in xs]   # (:20)
        grads_vs = sum(g * v for g, v in zip(grads, vs))                  # (:21)
        hess_vs = torch.autograd.grad(grads_vs, xs)                       # Hessian-vector products (:22)
        f_values.append(float(y.detach()))
        grads = [g.detach() for g in grads]
        Q = psgd.update_precond_dense(Q, vs, hess_vs, step=0.2)           # (:25)
        precond_grads = psgd.precond_grad_dense(Q, grads)                 # (:26)
        with torch.no_grad():
            for x, g in zip(xs, precond_grads):
                x.sub_(0.5 * g)                                           # (:27)
    return f_values, [float(x.detach()) for x in xs], Q


if __name__ == "__main__":
    f, xs, Q = run()
    print("f: %.3e -> %.3e after %d iterations; x = (%.6f, %.6f)" % (f[0], f[-1], len(f), xs[0], xs[1]))
