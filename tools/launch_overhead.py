"""CPU-side cost of one fused UVd step at a size where the GPU work is negligible (launch-bound regime)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import _lib

N, r = int(sys.argv[1]) if len(sys.argv) > 1 else 1021, 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
U = torch.randn(N, r, device=dev, generator=g) * (N * r) ** -0.5
V = torch.randn(N, r, device=dev, generator=g) * (N * r) ** -0.5
d = torch.ones(N, 1, device=dev)
gr, v = torch.randn(N, 1, device=dev, generator=g), torch.randn(N, 1, device=dev, generator=g)
h = v * 2.0
lib = _lib.load()
ws = psgd.uvd_workspace(dev, N, r)
out = torch.empty_like(gr)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()


def timed(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t_issue / n * 1e6, (time.perf_counter() - t0) / n * 1e6


calls = {
    "python wrapper (fused step)": lambda: psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, gr, 0.01, 1.1754944e-38, balance=False, update_U=True),
    "raw C ABI (fused step)": lambda: lib.psgd_uvd_update_apply_f32(P(U), P(V), P(d), P(v), P(h), P(gr), P(out), N, r, 0.01, 1.1754944e-38, 0, 1, P(ws), ws.numel(), st),
    "raw C ABI (apply)": lambda: lib.psgd_uvd_apply_f32(P(U), P(V), P(d), P(gr), P(out), N, r, P(ws), ws.numel(), st),
}
for name, fn in calls.items():
    a, b = timed(fn)
    print("%-30s N=%d  host issue %.1f us/call, end-to-end %.1f us/call" % (name, N, a, b), flush=True)
