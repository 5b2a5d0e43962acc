"""Every native sparse-LU rank (1 .. 64) at four row counts with partial last tiles: apply, update, apply on the updated factors against fp64
(the scan that isolated the r = 41 / 47 tail failure of round 5; prints only the failing lines when piped through grep -v).   python tools/r05_splu_rank_tail_scan.py"""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd
dev = torch.device("cuda")
def splu_apply64(L12, l3, U12, u3, g, r):
    L1, L2, U1, U2 = L12[:r], L12[r:], U12[:, :r], U12[:, r:]
    g1, g2 = g[:r], g[r:]
    Ug1 = U1 @ g1 + U2 @ g2
    Qg1 = L1 @ Ug1
    Qg2 = L2 @ Ug1 + l3 * (u3 * g2)
    Lt1 = L1.t() @ Qg1 + L2.t() @ Qg2
    return torch.cat([U1.t() @ Lt1, U2.t() @ Lt1 + u3 * (l3 * Qg2)], 0)
for r in list(range(1, 65)):
    for N in (r + 1, 257 + r, 9257, 30751):
        g = torch.Generator(device=dev).manual_seed(r * 7 + N)
        sc = 0.3 / r ** 0.5
        L12 = torch.randn(N, r, device=dev, generator=g) * (sc * 3 * (r / N) ** 0.5)
        U12 = torch.randn(r, N, device=dev, generator=g) * (sc * 3 * (r / N) ** 0.5)
        L12[:r] = torch.tril(torch.randn(r, r, device=dev, generator=g) * sc, -1) + torch.eye(r, device=dev)
        U12[:, :r] = torch.triu(torch.randn(r, r, device=dev, generator=g) * sc, 1) + torch.eye(r, device=dev)
        l3 = torch.exp(torch.empty(N - r, 1, device=dev).uniform_(-0.5, 0.5, generator=g))
        u3 = torch.exp(torch.empty(N - r, 1, device=dev).uniform_(-0.5, 0.5, generator=g)) * 0.7
        x = torch.randn(N, 1, device=dev, generator=g)
        dg = x * torch.exp(torch.empty(N, 1, device=dev).uniform_(-2, 2, generator=g))
        gr = torch.randn(N, 1, device=dev, generator=g)
        a = psgd.precond_grad_splu(L12, l3, U12, u3, [gr])[0]
        ref = splu_apply64(L12.double(), l3.double(), U12.double(), u3.double(), gr.double(), r)
        e1 = float((a.double() - ref).norm() / ref.norm())
        new = psgd.update_precond_splu(L12, l3, U12, u3, [x], [dg], 0.05)
        fin = [bool(torch.isfinite(t).all()) for t in new]
        a2 = psgd.precond_grad_splu(*new, [gr])[0]
        ref2 = splu_apply64(*[t.double() for t in new], gr.double(), r)
        e2 = float((a2.double() - ref2).norm() / ref2.norm())
        print("r=%d N=%d apply err %.2e  update finite %s  apply(new) err %.2e finite %s maxabs new %s" % (r, N, e1, fin, e2, bool(torch.isfinite(a2).all()), [float(t.abs().max()) for t in new]), flush=True)
