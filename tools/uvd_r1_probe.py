"""r = 1..3 UVd cases of the fuzz generator (V correlated with U, ||U V'|| = O(1)): error of the fused update+apply against the
fp64 restatement next to the conditioning of K = I + V'U (psgd.py:575), whose inverse the update applies twice."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fuzz_gpu as F

dev = F.dev
g = torch.Generator(device=dev).manual_seed(int(os.environ.get("SEED", "5")))
rows = []
for it in range(0, 9000, 3):                           # it % 3 == 0: the correlated, O(1) cases
    r = 1 + ((it // 3) % 3 if os.environ.get('RANKS', '1') == '3' else 0)
    N = int(torch.randint(max(r, 2), 20000, (1,), generator=g, device=dev))
    gain = float(torch.empty(1, device=dev).uniform_(0.5, 1.5, generator=g)) * r ** 0.5
    sc = gain * (1.0 / (N * r)) ** 0.5
    U, V = torch.randn(N, r, device=dev, generator=g) * sc, torch.randn(N, r, device=dev, generator=g) * sc
    if N > 4 * r:
        V = (0.5 * U @ torch.linalg.qr(torch.randn(r, r, device=dev, generator=g))[0] + 0.7 * V).contiguous()
    d = torch.exp(0.3 * torch.randn(N, 1, device=dev, generator=g))
    gr, v = torch.randn(N, 1, device=dev, generator=g), torch.randn(N, 1, device=dev, generator=g)
    h = v * torch.exp(torch.empty(N, 1, device=dev).uniform_(-4.6, 4.6, generator=g))
    K = torch.eye(r, device=dev, dtype=torch.float64) + V.double().t() @ U.double()
    cond = float(torch.linalg.cond(K)) if r > 1 else 1.0 / abs(float(K[0, 0]))      # r = 1: 1 / |K|
    U64, V64, d64 = U.double(), V.double(), d.double()
    U0, V0, d0 = U.clone(), V.clone(), d.clone()
    upd = bool(it % 2)
    out = F.psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, gr, 0.01, F.TINY, balance=False, update_U=upd)
    F.ref64.update_precond_UVd_math_(U64, V64, d64, v.double(), h.double(), 0.01, F.TINY, balance=False, update_U=upd)
    e = max(F.rel(out, F.ref64.precond_grad_UVd_math(U64, V64, d64, gr.double())), F.rel(U, U64), F.rel(V, V64), F.rel(d, d64))
    # empirical conditioning of the reference map itself: the same fp64 update on inputs perturbed by 1e-7 (relative)
    # cancellation in the normaliser of psgd.py:594-596 / :608-610 (fp64, from the inputs)
    Ud, Vd, dd = U0.double(), V0.double(), d0.double()
    t, w = dd * h.double(), v.double() / dd
    UU, VV, VU = Ud.t() @ Ud, Vd.t() @ Vd, Vd.t() @ Ud
    Ut, Uw, Vt, Vw = (Ud.t() @ t)[:, 0], (Ud.t() @ w)[:, 0], (Vd.t() @ t)[:, 0], (Vd.t() @ w)[:, 0]
    tt, tw, ww = float((t * t).sum()), float((t * w).sum()), float((w * w).sum())
    Kd = torch.eye(r, device=dev, dtype=torch.float64) + VU
    s1 = Vt
    x1 = torch.linalg.solve(Kd.t(), Uw)
    p2 = Vw - VV @ x1
    cs1 = VU @ s1
    aa = tt + 2 * (s1 @ Ut) + s1 @ (UU @ s1)
    bb = ww - 2 * (x1 @ Vw) + x1 @ (VV @ x1)
    ab = tw - x1 @ Vt + s1 @ Uw - x1 @ cs1
    e1, e2, Mm = (Vt + cs1, p2, VV) if upd else (Ut + UU @ s1, Uw - VU.t() @ x1, UU)
    T1, T2, T3 = aa * (e1 @ (Mm @ e1)), bb * (e2 @ (Mm @ e2)), 2 * ab * (e1 @ (Mm @ e2))
    kappa = float((abs(T1) + abs(T2) + abs(T3)) / abs(T1 + T2 - T3))
    sens = kappa
    if e > 8e-7:                                      # the GPU's Gram of [U | V | t | w] against fp64, entry by entry
        from psgd_tf_amd import uvd_wide, preconditioned_stochastic_gradient_descent as impl
        cx = uvd_wide._Ctx(U0, impl.uvd_workspace)
        Gg = cx.gram_pair(U0.contiguous(), V0.contiguous(), d0.reshape(-1), v.reshape(-1), h.reshape(-1))
        W = torch.cat([Ud, Vd, t, w], 1)
        G64 = W.t() @ W
        nrm_ = torch.sqrt(torch.outer(G64.diagonal(), G64.diagonal()))
        print("N=%d: Gram error / sqrt(G_ii G_jj):" % N, ((Gg - G64).abs() / nrm_).cpu().numpy().round(10).tolist(),
              " relative:", ((Gg - G64).abs() / G64.abs()).cpu().numpy().round(9).tolist(), flush=True)
    which = max((F.rel(U, U64), "U"), (F.rel(V, V64), "V"), (F.rel(d, d64), "d"))[1]
    rows.append((e, cond, r, N, sens, which))
rows.sort(reverse=True)
print("worst ten: err, cond(K) (r = 1: 1 / |K|), r, N")
for e, c, r, N, sens, which in rows[:10]:
    print("%.2e  cond %.1f  r=%d N=%d   worst output %s, cancellation in the normaliser of :594-596 (sum of |terms| / |sum|) %.0f" % (e, c, r, N, which, sens))
ratio = sorted(x[0] / x[4] for x in rows)
print("cases", len(rows), "| err / cancellation factor: median %.1e, max %.1e" % (ratio[len(ratio) // 2], ratio[-1]))
