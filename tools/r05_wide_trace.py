"""One wide-rank UVd update / apply / fused step in a loop for rocprofv3 --kernel-trace --stats:  python tools/r05_wide_trace.py N r [update|apply|step] [reps]"""
import sys
import torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
N, r = int(sys.argv[1]), int(sys.argv[2])
what = sys.argv[3] if len(sys.argv) > 3 else "update"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(r)
sc = (1.0 / (N * r)) ** 0.5
U, V = torch.randn(N, r, device=dev, generator=g) * sc, torch.randn(N, r, device=dev, generator=g) * sc
d = torch.ones(N, 1, device=dev)
gr, v = torch.randn(N, 1, device=dev, generator=g), torch.randn(N, 1, device=dev, generator=g)
h = v * 1.5
for i in range(reps):
    if what == "update":
        psgd.update_precond_UVd_math_(U, V, d, v, h, 0.01, 1e-38, balance=False, update_U=(i % 2 == 0))
    elif what == "step":
        psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, gr, 0.01, 1e-38, balance=False, update_U=(i % 2 == 0))
    else:
        psgd.precond_grad_UVd_math(U, V, d, gr)
torch.cuda.synchronize()
