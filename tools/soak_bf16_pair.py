"""Soak test of the in-launch hand-offs of the fused triangular pair (bf16 Kron apply): many back-to-back applies on
several shapes with changing gradients; every result is compared with the staged chain (no hand-offs) and the bounded
spins must never time out.   python tools/soak_bf16_pair.py [seconds]"""
import sys
import time
import torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib, kron  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)


def tri(n):
    return torch.triu(torch.randn(n, n, device=dev, generator=g) * (0.5 / n ** 0.5), 1) + torch.diag(torch.exp(0.3 * torch.randn(n, device=dev, generator=g)))


shapes = [(4096, 4096), (2048, 4096), (4096, 2048), (3072, 3072), (8192, 4096), (2560, 5120)]
state = {s: (tri(s[0]), tri(s[1])) for s in shapes}
t0, n, worst, timeouts = time.time(), 0, 0.0, 0
while time.time() - t0 < budget:
    M, N = shapes[n % len(shapes)]
    Ql, Qr = state[(M, N)]
    G = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    outs = [psgd.precond_grad_kron(Ql, Qr, G) for _ in range(8)]          # back to back, same operands
    torch.cuda.synchronize()
    timeouts = kron.check_bf16_handoffs()                                 # recoveries so far (results are unaffected)
    lib.psgd_kron_bf16_set_tuning(0, 1)                                   # staged chain on the 128^2 kernel: no hand-offs
    ref = psgd.precond_grad_kron(Ql, Qr, G)
    lib.psgd_kron_bf16_set_tuning(0, 0)
    for o in outs:
        assert torch.equal(o, outs[0]), "hand-off result changed between identical launches (%d x %d)" % (M, N)
    e = ((outs[0].float() - ref.float()).norm() / ref.float().norm()).item()
    worst = max(worst, e)
    assert e < 1e-2, (M, N, e)
    n += 1
print("rounds %d (8 applies each), hand-off timeouts %d, worst rel diff vs the staged chain %.2e" % (n, timeouts, worst))
sys.exit(1 if timeouts else 0)
