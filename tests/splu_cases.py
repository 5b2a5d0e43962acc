"""Seeded synthetic sparse-LU (splu) preconditioner problems shared by the CPU and GPU tests.

Layout is the reference's (psgd.py:396-404): L12 = [L1; L2] is [N, r] with L1 lower triangular,
U12 = [U1, U2] is [r, N] with U1 upper triangular, l3 and u3 are [N - r, 1]."""
import numpy as np


def make_splu_problem(N, r, seed=0, gain=0.3, diag_spread=0.5, init_like_demo=False):
    """init_like_demo: the state of demo_usage_of_all_preconditioners.py:47-51 (0.1*I corners, zero
    L2/U2, 0.1 diagonals); otherwise a generic well-conditioned state.  dx ~ N(0,1), dg = c .* dx
    with c ~ LogUniform[1e-1, 1e1] (a diagonal SPD Hessian), g ~ N(0,1).  fp32 arrays."""
    rng = np.random.default_rng(seed)
    n2 = N - r
    if init_like_demo:
        L12 = 0.1 * np.concatenate([np.eye(r), np.zeros((n2, r))], 0)
        U12 = 0.1 * np.concatenate([np.eye(r), np.zeros((r, n2))], 1)
        l3 = 0.1 * np.ones((n2, 1))
        u3 = 0.1 * np.ones((n2, 1))
    else:
        sc = gain / np.sqrt(r)
        L1 = np.tril(rng.standard_normal((r, r)) * sc, -1) + np.diag(np.exp(diag_spread * rng.uniform(-1, 1, r)))
        U1 = np.triu(rng.standard_normal((r, r)) * sc, 1) + np.diag(np.exp(diag_spread * rng.uniform(-1, 1, r)))
        L12 = np.concatenate([L1, rng.standard_normal((n2, r)) * sc * np.sqrt(r / max(N, 1)) * 3], 0)
        U12 = np.concatenate([U1, rng.standard_normal((r, n2)) * sc * np.sqrt(r / max(N, 1)) * 3], 1)
        l3 = np.exp(diag_spread * rng.uniform(-1, 1, (n2, 1)))
        u3 = np.exp(diag_spread * rng.uniform(-1, 1, (n2, 1))) * 0.7
    dx = rng.standard_normal((N, 1))
    dg = np.exp(rng.uniform(np.log(1e-1), np.log(1e1), (N, 1))) * dx
    g = rng.standard_normal((N, 1))
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(L12=f(L12), l3=f(l3), U12=f(U12), u3=f(u3), dx=f(dx), dg=f(dg), g=f(g))


def splu_dense(L12, l3, U12, u3):
    """The N x N factors L = [L1 0; L2 diag(l3)], U = [U1 U2; 0 diag(u3)] (psgd.py:399-403)."""
    N, r = L12.shape
    L = np.zeros((N, N), dtype=L12.dtype)
    U = np.zeros((N, N), dtype=L12.dtype)
    L[:, :r] = L12
    U[:r, :] = U12
    idx = np.arange(r, N)
    L[idx, idx] = l3[:, 0]
    U[idx, idx] = u3[:, 0]
    return L, U
