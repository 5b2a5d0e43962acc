"""The Kronecker-product preconditioner on the LeNet5 weight shapes of the reference's mnist_with_lenet5.py (:12-16), through the
reference's own call pattern (:51, :53) -- one update_precond_kron / precond_grad_kron call per layer -- eager and captured once in
a CUDA graph (the five Python calls of a step then cost the host nothing; the calls are the same, the graph is replayed).
The layers are independent of each other, and a small layer's call is a chain of 3-5 dependent launches, so the third form
wraps each list comprehension in `with kron.layer_batch():` -- the calls inside only queue their work and return their (not yet
filled) outputs; leaving the block issues ONE batched launch sequence for the whole layer set (5 launches for the five updates,
3 for the five applies).  (`kron.layer_streams()` is the other one-line form: every call on its own forked stream.)

Synthetic data: there is no MNIST here.  A step = preconditioner update on a (dX, dG) pair with dG = Hl dX Hr for fixed SPD
Hl, Hr (what a quadratic loss would give), then the preconditioned gradient.  After a few hundred steps Ql'Ql (x) Qr'Qr has
whitened Hl (x) Hr: the script prints how far the preconditioned Hessian is from the identity, and the time per step.

    python examples/lenet5_kron_step.py [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import kron  # noqa: E402
import contextlib  # noqa: E402

SHAPES = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10)]          # W1 .. W5 of mnist_with_lenet5.py:12-16


def main(steps=300):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    Qs = [[torch.eye(m, device=dev), torch.eye(n, device=dev)] for m, n in SHAPES]                  # :61-62
    Hl = [torch.diag(torch.exp(torch.empty(m, device=dev).uniform_(-1.5, 1.5, generator=g))) for m, n in SHAPES]
    Hr = [torch.diag(torch.exp(torch.empty(n, device=dev).uniform_(-1.5, 1.5, generator=g))) for m, n in SHAPES]
    dXs = [torch.empty(m, n, device=dev) for m, n in SHAPES]
    dGs = [torch.empty(m, n, device=dev) for m, n in SHAPES]
    Gs = [torch.randn(m, n, device=dev, generator=g) for m, n in SHAPES]
    pre = [torch.empty_like(x) for x in Gs]

    def draw():
        for i in range(len(SHAPES)):
            dXs[i].normal_(generator=g)
            torch.matmul(torch.matmul(Hl[i], dXs[i]), Hr[i], out=dGs[i])

    def step(batched=False):                                              # the reference's two list comprehensions
        block = kron.layer_batch if batched else contextlib.nullcontext
        with block():
            new = [psgd.update_precond_kron(ql, qr, dx, dg, 0.05) for (ql, qr), dx, dg in zip(Qs, dXs, dGs)]      # :51
        for q, (a, b) in zip(Qs, new):
            q[0].copy_(a); q[1].copy_(b)                                  # (static buffers: the graph is replayed on them)
        with block():
            out = [psgd.precond_grad_kron(ql, qr, gr) for (ql, qr), gr in zip(Qs, Gs)]                            # :53
        for p, o in zip(pre, out):
            p.copy_(o)

    draw()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    eager_us = (time.perf_counter() - t0) / 50 * 1e6
    side = torch.cuda.Stream()
    graphs = []
    for batched in (False, True):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(batched)                                                 # (warm-up on the capture stream: workspaces exist before the capture)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            step(batched)
        graphs.append(graph)
    t0 = time.perf_counter()
    for _ in range(50):
        step(True)
    torch.cuda.synchronize()
    eager_batch_us = (time.perf_counter() - t0) / 50 * 1e6
    t_graph = [0.0, 0.0]
    for it in range(steps):
        draw()
        which = it & 1                                                    # both graphs advance the same factors: alternate
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        graphs[which].replay()
        torch.cuda.synchronize()
        t_graph[which] += time.perf_counter() - t0
    spread = lambda a: float(torch.linalg.matrix_norm(a / (torch.trace(a) / a.shape[0]) - torch.eye(a.shape[0], device=dev), ord=2))
    worst0 = max(max(spread(hl), spread(hr)) for hl, hr in zip(Hl, Hr))   # the unpreconditioned Hessian factors
    worst = 0.0
    for (ql, qr), hl, hr in zip(Qs, Hl, Hr):                              # P H = (Qr'Qr (x) Ql'Ql)(Hr (x) Hl) -> c I
        pl, pr = ql.t() @ ql @ hl, qr.t() @ qr @ hr
        c = (torch.trace(pl) / pl.shape[0]) * (torch.trace(pr) / pr.shape[0])
        dev_l = torch.linalg.matrix_norm(pl / (torch.trace(pl) / pl.shape[0]) - torch.eye(pl.shape[0], device=dev), ord=2)
        dev_r = torch.linalg.matrix_norm(pr / (torch.trace(pr) / pr.shape[0]) - torch.eye(pr.shape[0], device=dev), ord=2)
        worst = max(worst, float(dev_l), float(dev_r))
        assert torch.isfinite(c)
    half = max(steps // 2, 1)
    print("LeNet5 layer set, per-layer calls: eager %.0f us per step (update + apply), graph replay %.0f us; inside kron.layer_batch "
          "blocks: eager %.0f us, graph replay %.0f us; max ||P H / mean - I||_2 over the ten factors: %.3f after %d steps "
          "(%.3f without a preconditioner)" % (eager_us, t_graph[0] / (steps - steps // 2) * 1e6, eager_batch_us,
                                               t_graph[1] / half * 1e6, worst, steps, worst0))
    return worst0, worst


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
