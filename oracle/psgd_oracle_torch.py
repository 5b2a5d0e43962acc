"""Multi-threaded CPU restatement of the reference op sequence on torch-CPU.  TEST / BASELINE
INFRASTRUCTURE ONLY (same rules as psgd_oracle.py: never imported by the product).

Purpose: the `cpu_baseline` leg of bench.py, and (device-agnostic torch ops, so it also runs in fp64 on the
GPU through rocBLAS) the independent full-size cross-check of tests/test_full_size_gpu.py.  The reference is TensorFlow-eager Python
(psgd.py); TensorFlow is not available, so the baseline is this restatement: one torch library
call per TF op, same association order and same temporaries as psgd.py:540-627 (UVd) and
psgd.py:156-192 (Kron dense(x)dense), run with torch's intra-op thread pool on the host cores.
PARITY UNPINNED in the same sense as psgd_oracle.py; tests/test_oracle_kat.py checks it
against the NumPy oracle.
"""
import torch


def IpUVtmatvec(U, V, x):
    """psgd.py:540-544."""
    return x + torch.matmul(U, torch.matmul(V.t(), x))


def precond_grad_UVd_math(U, V, d, g):
    """psgd.py:619-627."""
    g = IpUVtmatvec(U, V, d * g)
    return d * IpUVtmatvec(V, U, g)


def update_precond_UVd_math_(U, V, d, v, h, step, tiny, balance=False, update_U=True):
    """psgd.py:554-617 (in place on U or V, and d)."""
    if balance:
        rho = torch.sqrt(torch.max(torch.abs(U)) / torch.max(torch.abs(V)))
        U.copy_(U / rho)
        V.copy_(rho * V)
    Qh = IpUVtmatvec(U, V, d * h)
    Ph = d * IpUVtmatvec(V, U, Qh)
    VtU = torch.matmul(V.t(), U)
    IpVtU = torch.eye(VtU.shape[0], dtype=VtU.dtype, device=VtU.device) + VtU
    invQtv = v / d
    invQtv = invQtv - torch.matmul(V, torch.linalg.solve(IpVtU.t(), torch.matmul(U.t(), invQtv)))
    invPv = invQtv - torch.matmul(U, torch.linalg.solve(IpVtU, torch.matmul(V.t(), invQtv)))
    invPv = invPv / d
    nablaD = Ph * h - v * invPv
    mu = step / (torch.max(torch.abs(nablaD)) + tiny)
    a, b = Qh, invQtv
    d.sub_(mu * d * nablaD)
    if update_U:
        atV = torch.matmul(a.t(), V)
        atVVt = torch.matmul(atV, V.t())
        btV = torch.matmul(b.t(), V)
        btVVt = torch.matmul(btV, V.t())
        norm = torch.sqrt(torch.abs(torch.matmul(a.t(), a) * torch.matmul(atVVt, atVVt.t())
                                    + torch.matmul(b.t(), b) * torch.matmul(btVVt, btVVt.t())
                                    - 2 * torch.matmul(a.t(), b) * torch.matmul(atVVt, btVVt.t())))
        mu = step / (norm + tiny)
        U.sub_(mu * (torch.matmul(a, torch.matmul(atV, IpVtU)) - torch.matmul(b, torch.matmul(btV, IpVtU))))
    else:
        atU = torch.matmul(a.t(), U)
        btU = torch.matmul(b.t(), U)
        UUta = torch.matmul(U, atU.t())
        UUtb = torch.matmul(U, btU.t())
        norm = torch.sqrt(torch.abs(torch.matmul(UUta.t(), UUta) * torch.matmul(a.t(), a)
                                    + torch.matmul(UUtb.t(), UUtb) * torch.matmul(b.t(), b)
                                    - 2 * torch.matmul(UUta.t(), UUtb) * torch.matmul(a.t(), b)))
        mu = step / (norm + tiny)
        V.sub_(mu * (torch.matmul(a + torch.matmul(V, atU.t()), atU) - torch.matmul(b + torch.matmul(V, btU.t()), btU)))
    return None


def _solve_ut_adjoint(Q, X):
    return torch.linalg.solve_triangular(Q.t(), X, upper=False)


def update_precond_dense_dense(Ql, Qr, dX, dG, step, tiny):
    """psgd.py:156-179."""
    rho = torch.sqrt(torch.max(torch.diagonal(Ql)) / torch.max(torch.diagonal(Qr)))
    Ql = Ql / rho
    Qr = rho * Qr
    A = torch.matmul(Ql, torch.matmul(dG, Qr.t()))
    Bt = _solve_ut_adjoint(Ql, _solve_ut_adjoint(Qr, dX.t()).t())
    grad1 = torch.triu(torch.matmul(A, A.t()) - torch.matmul(Bt, Bt.t()))
    grad2 = torch.triu(torch.matmul(A.t(), A) - torch.matmul(Bt.t(), Bt))
    step1 = step / (torch.max(torch.abs(grad1)) + tiny)
    step2 = step / (torch.max(torch.abs(grad2)) + tiny)
    return Ql - torch.matmul(step1 * grad1, Ql), Qr - torch.matmul(step2 * grad2, Qr)


def precond_grad_dense_dense(Ql, Qr, Grad):
    """psgd.py:182-192."""
    if Grad.shape[0] < Grad.shape[1]:
        return torch.matmul(torch.matmul(torch.matmul(torch.matmul(Ql.t(), Ql), Grad), Qr.t()), Qr)
    return torch.matmul(Ql.t(), torch.matmul(Ql, torch.matmul(Grad, torch.matmul(Qr.t(), Qr))))
