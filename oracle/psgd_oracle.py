"""CPU oracle for the PSGD preconditioner hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the arithmetic in the reference module
``preconditioned_stochastic_gradient_descent.py`` (called ``psgd.py`` below).
It exists so that the HIP kernels have something to be checked against; it is
never imported by the product package (``psgd_tf_amd``).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.

PARITY UNPINNED: the reference is Python-on-TensorFlow, TensorFlow is not
installable in the build container, and the reference ships no tests, golden
vectors or fixtures.  What pins this restatement instead:
  * analytic known-answer tests derived from the source alone
    (tests/test_oracle_kat.py: identity, fixed point, Woodbury inverse,
    Rosenbrock first step, dispatcher table, triangularity ...);
  * cross-checks against independent dense linear algebra in fp64.
These pin the *mathematical specification as read from source*, not
TensorFlow's rounding.

Conventions
  * Every function works in the dtype of its inputs (fp32 or fp64) and keeps
    the reference's association order, so the fp32 run is the "reference op
    sequence" and the fp64 run is the ground truth.
  * Random control flow of the reference (psgd.py:562, :588) is exposed as the
    explicit arguments ``balance`` and ``update_U``.
  * ``update_precond_UVd_math_`` mutates U, V, d in place like the reference
    (psgd.py:566-567,584,600,614); everything else is pure.
"""
import numpy as np
import scipy.linalg as sla


# --------------------------------------------------------------------------- constants
def tiny_of(dtype):
    """psgd.py:22 / :682 -- smallest *normal* positive number of dtype.

    The reference finds it by recursive halving under TF's flush-to-zero
    arithmetic; without FTZ that loop would give the smallest subnormal, so the
    value is taken from finfo instead."""
    return np.finfo(np.dtype(dtype)).tiny


def delta_param_scale_of(dtype):
    """psgd.py:683 -- sqrt(machine eps): halving while 1 + x/2 > 1, then **0.5."""
    dt = np.dtype(dtype).type
    return dt(np.finfo(np.dtype(dtype)).eps) ** dt(0.5)


def _col(x):
    return np.reshape(x, (-1, 1))


def _utri_solve_adjoint(Q, X):
    """tf.linalg.triangular_solve(Q, X, lower=False, adjoint=True): solve Q^T Y = X
    using only the upper triangle of Q (psgd.py:39,174)."""
    return sla.solve_triangular(Q, X, lower=False, trans='T', check_finite=False).astype(X.dtype, copy=False)


def _triu(X):
    """tf.linalg.band_part(X, 0, -1) (psgd.py:40,175-176)."""
    return np.triu(X)


# --------------------------------------------------------------------------- dense (plumbing)
def update_precond_dense(Q, dxs, dgs, step=0.01):
    """psgd.py:26-42."""
    dt = Q.dtype.type
    dx = np.concatenate([_col(x) for x in dxs], 0).astype(Q.dtype, copy=False)   # :34
    dg = np.concatenate([_col(g) for g in dgs], 0).astype(Q.dtype, copy=False)   # :35
    a = Q @ dg                                                                   # :38
    b = _utri_solve_adjoint(Q, dx)                                               # :39
    grad = _triu(a @ a.T - b @ b.T)                                              # :40
    step0 = dt(step) / (np.max(np.abs(grad)) + tiny_of(Q.dtype))                 # :41
    return Q - (step0 * grad) @ Q                                                # :42


def precond_grad_dense(Q, grads):
    """psgd.py:45-63: list in, list out with the original shapes."""
    cols = [_col(np.asarray(g, dtype=Q.dtype)) for g in grads]
    lens = [c.shape[0] for c in cols]
    g = np.concatenate(cols, 0)
    pg = Q.T @ (Q @ g)                                                           # :55
    out, idx = [], 0
    for g0, n in zip(grads, lens):                                               # :57-61
        out.append(np.reshape(pg[idx:idx + n], np.shape(g0)))
        idx += n
    return out


# --------------------------------------------------------------------------- Kron: dense (x) dense
def update_precond_dense_dense(Ql, Qr, dX, dG, step=0.01):
    """psgd.py:156-179 (K0..K6 of SURVEY 2.3)."""
    dt = Ql.dtype.type
    tiny = tiny_of(Ql.dtype)
    max_l = np.max(np.diag(Ql))                                                  # :166
    max_r = np.max(np.diag(Qr))                                                  # :167
    rho = np.sqrt(max_l / max_r)                                                 # :168
    Ql = Ql / rho                                                                # :169
    Qr = rho * Qr                                                                # :170
    A = Ql @ (dG @ Qr.T)                                                         # :173
    Bt = _utri_solve_adjoint(Ql, _utri_solve_adjoint(Qr, dX.T).T)                # :174
    grad1 = _triu(A @ A.T - Bt @ Bt.T)                                           # :175
    grad2 = _triu(A.T @ A - Bt.T @ Bt)                                           # :176
    step1 = dt(step) / (np.max(np.abs(grad1)) + tiny)                            # :177
    step2 = dt(step) / (np.max(np.abs(grad2)) + tiny)                            # :178
    return Ql - (step1 * grad1) @ Ql, Qr - (step2 * grad2) @ Qr                  # :179


def precond_grad_dense_dense(Ql, Qr, Grad):
    """psgd.py:182-192: association order depends on the strict test M < N."""
    if Grad.shape[0] < Grad.shape[1]:                                            # :189
        return (((Ql.T @ Ql) @ Grad) @ Qr.T) @ Qr                                # :190
    return Ql.T @ (Ql @ (Grad @ (Qr.T @ Qr)))                                    # :192


# --------------------------------------------------------------------------- Kron: normalization (x) dense
def _ql_times(ql, X):
    """Ql*X for the normalization format (psgd.py:218-219, 258-259): Ql has diagonal
    ql[0] and last column ql[1] (its last entry unused in the product)."""
    return ql[0:1].T * X + ql[1:].T @ X[-1:]


def _ql_inv_t_times(ql, X):
    """Ql^(-T)*X for the normalization format (psgd.py:230-232, 353-355)."""
    Bt = (1.0 / ql[0:1]).T.astype(X.dtype) * X
    last = Bt[-1:] - (ql[1:] / (ql[0:1] * ql[0, -1])) @ X
    return np.concatenate([Bt[:-1], last], axis=0)


def _norm_grad1(A, Bt):
    """psgd.py:235-237 / :358-360."""
    g_diag = np.sum(A * A, axis=1) - np.sum(Bt * Bt, axis=1)
    g_bias = A[:-1] @ A[-1:].T - Bt[:-1] @ Bt[-1:].T
    g_bias = np.concatenate([np.reshape(g_bias, (-1,)), np.zeros(1, A.dtype)], axis=0)
    return g_diag, g_bias


def update_precond_norm_dense(ql, Qr, dX, dG, step=0.01):
    """psgd.py:198-246."""
    dt = Qr.dtype.type
    tiny = tiny_of(Qr.dtype)
    rho = np.sqrt(np.max(ql[0]) / np.max(np.diag(Qr)))                           # :211-213
    ql = ql / rho
    Qr = rho * Qr
    A = _ql_times(ql, dG) @ Qr.T                                                 # :218-220
    Bt = _utri_solve_adjoint(Qr, _ql_inv_t_times(ql, dX).T).T                    # :230-233
    g_diag, g_bias = _norm_grad1(A, Bt)                                          # :235-237
    step1 = dt(step) / (max(np.max(np.abs(g_diag)), np.max(np.abs(g_bias))) + tiny)   # :239
    new_ql0 = ql[0] - step1 * g_diag * ql[0]                                     # :240
    new_ql1 = ql[1] - step1 * (g_diag * ql[1] + ql[0, -1] * g_bias)              # :241
    grad2 = _triu(A.T @ A - Bt.T @ Bt)                                           # :243
    step2 = dt(step) / (np.max(np.abs(grad2)) + tiny)                            # :244
    return np.stack((new_ql0, new_ql1)), Qr - (step2 * grad2) @ Qr               # :246


def _norm_left_gram_apply(ql, preG):
    """Ql^T * preG for the normalization format (psgd.py:265-268, 386-389)."""
    add_last_row = ql[1:] @ preG
    preG = ql[0:1].T * preG
    return np.concatenate([preG[:-1], preG[-1:] + add_last_row], axis=0)


def precond_grad_norm_dense(ql, Qr, Grad):
    """psgd.py:249-270."""
    preG = _ql_times(ql, Grad)                                                   # :258-259
    if preG.shape[0] < preG.shape[1]:                                            # :260
        preG = (preG @ Qr.T) @ Qr
    else:
        preG = preG @ (Qr.T @ Qr)
    return _norm_left_gram_apply(ql, preG)


# --------------------------------------------------------------------------- Kron: dense (x) scaling
def update_precond_dense_scale(Ql, qr, dX, dG, step=0.01):
    """psgd.py:276-307."""
    dt = Ql.dtype.type
    tiny = tiny_of(Ql.dtype)
    rho = np.sqrt(np.max(np.diag(Ql)) / np.max(qr))                              # :288-290
    Ql = Ql / rho
    qr = rho * qr
    A = (Ql @ dG) * qr                                                           # :295-296
    Bt = _utri_solve_adjoint(Ql, dX) * (1.0 / qr)                                # :298-299
    grad1 = _triu(A @ A.T - Bt @ Bt.T)                                           # :301
    step1 = dt(step) / (np.max(np.abs(grad1)) + tiny)
    grad2 = np.sum(A * A, axis=0, keepdims=True) - np.sum(Bt * Bt, axis=0, keepdims=True)   # :304
    step2 = dt(step) / (np.max(np.abs(grad2)) + tiny)
    return Ql - (step1 * grad1) @ Ql, qr - step2 * grad2 * qr                    # :307


def precond_grad_dense_scale(Ql, qr, Grad):
    """psgd.py:310-322."""
    if Grad.shape[0] < Grad.shape[1]:
        preG = (Ql.T @ Ql) @ Grad
    else:
        preG = Ql.T @ (Ql @ Grad)
    return preG * (qr * qr)


# --------------------------------------------------------------------------- Kron: normalization (x) scaling
def update_precond_norm_scale(ql, qr, dX, dG, step=0.01):
    """psgd.py:328-369."""
    dt = qr.dtype.type
    tiny = tiny_of(qr.dtype)
    rho = np.sqrt(np.max(ql[0]) / np.max(qr))                                    # :342-344
    ql = ql / rho
    qr = rho * qr
    A = _ql_times(ql, dG) * qr                                                   # :349-351
    Bt = _ql_inv_t_times(ql, dX) * (1.0 / qr)                                    # :353-356
    g_diag, g_bias = _norm_grad1(A, Bt)
    step1 = dt(step) / (max(np.max(np.abs(g_diag)), np.max(np.abs(g_bias))) + tiny)
    new_ql0 = ql[0] - step1 * g_diag * ql[0]
    new_ql1 = ql[1] - step1 * (g_diag * ql[1] + ql[0, -1] * g_bias)
    grad2 = np.sum(A * A, axis=0, keepdims=True) - np.sum(Bt * Bt, axis=0, keepdims=True)
    step2 = dt(step) / (np.max(np.abs(grad2)) + tiny)
    return np.stack((new_ql0, new_ql1)), qr - step2 * grad2 * qr


def precond_grad_norm_scale(ql, qr, Grad):
    """psgd.py:372-391."""
    preG = _ql_times(ql, Grad) * (qr * qr)
    return _norm_left_gram_apply(ql, preG)


# --------------------------------------------------------------------------- Kron dispatchers
KRON_FORMATS = ('dense_dense', 'dense_norm', 'dense_scale', 'norm_dense',
                'norm_scale', 'scale_dense', 'scale_norm', 'unknown')


def kron_format(shape_l, shape_r):
    """Shape dispatch of psgd.py:80-110 / :122-152.  Square is
    tested first, so [1,1] and [2,2] factors are dense."""
    m, n = shape_l
    p, q = shape_r
    if m == n:
        if p == q:
            return 'dense_dense'
        if p == 2:
            return 'dense_norm'
        if p == 1:
            return 'dense_scale'
        return 'unknown'
    if m == 2:
        if p == q:
            return 'norm_dense'
        if p == 1:
            return 'norm_scale'
        return 'unknown'
    if m == 1:
        if p == q:
            return 'scale_dense'
        if p == 2:
            return 'scale_norm'
        return 'unknown'
    return 'unknown'


def update_precond_kron(Ql, Qr, dX, dG, step=0.01):
    """psgd.py:72-110.  Unknown format: inputs returned unchanged (:89-91 etc.)."""
    fmt = kron_format(Ql.shape, Qr.shape)
    if fmt == 'dense_dense':
        return update_precond_dense_dense(Ql, Qr, dX, dG, step)                  # :84
    if fmt == 'dense_norm':
        return update_precond_norm_dense(Qr, Ql, dX.T, dG.T, step)[::-1]         # :86
    if fmt == 'dense_scale':
        return update_precond_dense_scale(Ql, Qr, dX, dG, step)                  # :88
    if fmt == 'norm_dense':
        return update_precond_norm_dense(Ql, Qr, dX, dG, step)                   # :94
    if fmt == 'norm_scale':
        return update_precond_norm_scale(Ql, Qr, dX, dG, step)                   # :96
    if fmt == 'scale_dense':
        return update_precond_dense_scale(Qr, Ql, dX.T, dG.T, step)[::-1]        # :102
    if fmt == 'scale_norm':
        return update_precond_norm_scale(Qr, Ql, dX.T, dG.T, step)[::-1]         # :104
    return Ql, Qr


def precond_grad_kron(Ql, Qr, Grad):
    """psgd.py:116-152.  Unknown format: Grad returned unchanged."""
    fmt = kron_format(Ql.shape, Qr.shape)
    if fmt == 'dense_dense':
        return precond_grad_dense_dense(Ql, Qr, Grad)                            # :126
    if fmt == 'dense_norm':
        return precond_grad_norm_dense(Qr, Ql, Grad.T).T                         # :128
    if fmt == 'dense_scale':
        return precond_grad_dense_scale(Ql, Qr, Grad)                            # :130
    if fmt == 'norm_dense':
        return precond_grad_norm_dense(Ql, Qr, Grad)                             # :136
    if fmt == 'norm_scale':
        return precond_grad_norm_scale(Ql, Qr, Grad)                             # :138
    if fmt == 'scale_dense':
        return precond_grad_dense_scale(Qr, Ql, Grad.T).T                        # :144
    if fmt == 'scale_norm':
        return precond_grad_norm_scale(Qr, Ql, Grad.T).T                         # :146
    return Grad


# --------------------------------------------------------------------------- sparse LU (splu)
def _splu_blocks(L12, U12):
    """psgd.py:420-424 / :499-503."""
    r = U12.shape[0]
    return r, L12[:r], L12[r:], U12[:, :r], U12[:, r:]


def _tri_solve(A, b, lower, adjoint=False):
    """tf.linalg.triangular_solve(A, b, lower=..., adjoint=...) (psgd.py:436,440,448,452)."""
    return sla.solve_triangular(A, b, lower=lower, trans='T' if adjoint else 'N',
                                check_finite=False).astype(A.dtype)


def update_precond_splu(L12, l3, U12, u3, dxs, dgs, step=0.01):
    """psgd.py:396-480.  Q = L U, L = [L1 0; L2 diag(l3)], U = [U1 U2; 0 diag(u3)].  Pure."""
    dt = L12.dtype.type
    step = dt(step)
    tiny = tiny_of(L12.dtype)
    r0 = U12.shape[0]
    # signed max, as written; tf.reduce_max of an empty tensor (N == r) is -inf
    max_l = max(np.max(np.diag(L12[:r0])), np.max(l3, initial=-np.inf))            # :411
    max_u = max(np.max(np.diag(U12[:, :r0])), np.max(u3, initial=-np.inf))         # :412
    rho = np.sqrt(max_l / max_u)                                                   # :413
    L12 = L12 / rho                                                                # :414-417
    l3 = l3 / rho
    U12 = rho * U12
    u3 = rho * u3

    r, L1, L2, U1, U2 = _splu_blocks(L12, U12)                                     # :420-424
    dx = np.concatenate([np.reshape(x, (-1, 1)) for x in dxs], 0).astype(L12.dtype)   # :426
    dg = np.concatenate([np.reshape(g, (-1, 1)) for g in dgs], 0).astype(L12.dtype)   # :427

    Ug1 = U1 @ dg[:r] + U2 @ dg[r:]                                                # :430
    Ug2 = u3 * dg[r:]                                                              # :431
    Qg1 = L1 @ Ug1                                                                 # :433
    Qg2 = L2 @ Ug1 + l3 * Ug2                                                      # :434
    iUtx1 = _tri_solve(U1, dx[:r], lower=False, adjoint=True)                      # :436
    iUtx2 = (dx[r:] - U2.T @ iUtx1) / u3                                           # :437
    iQtx2 = iUtx2 / l3                                                             # :439
    iQtx1 = _tri_solve(L1, iUtx1 - L2.T @ iQtx2, lower=True, adjoint=True)         # :440
    LtQg1 = L1.T @ Qg1 + L2.T @ Qg2                                                # :442
    LtQg2 = l3 * Qg2                                                               # :443
    Pg1 = U1.T @ LtQg1                                                             # :445
    Pg2 = U2.T @ LtQg1 + u3 * LtQg2                                                # :446
    iLiQtx1 = _tri_solve(L1, iQtx1, lower=True)                                    # :448
    iLiQtx2 = (iQtx2 - L2 @ iLiQtx1) / l3                                          # :449
    iPx2 = iLiQtx2 / u3                                                            # :451
    iPx1 = _tri_solve(U1, iLiQtx1 - U2 @ iPx2, lower=False)                        # :452

    grad1 = np.tril(Qg1 @ Qg1.T - iQtx1 @ iQtx1.T)                                 # :455-456
    grad2 = Qg2 @ Qg1.T - iQtx2 @ iQtx1.T                                          # :457
    grad3 = Qg2 * Qg2 - iQtx2 * iQtx2                                              # :458
    max_abs_grad = max(np.max(np.abs(grad1)), _max0(np.abs(grad2)), _max0(np.abs(grad3)))   # :459-461
    step0 = step / (max_abs_grad + tiny)                                           # :462
    newL1 = L1 - (step0 * grad1) @ L1                                              # :463
    newL2 = L2 - (step0 * grad2) @ L1 - step0 * grad3 * L2                         # :464
    newl3 = l3 - step0 * grad3 * l3                                                # :465

    grad1 = np.triu(Pg1 @ dg[:r].T - dx[:r] @ iPx1.T)                              # :468-469
    grad2 = Pg1 @ dg[r:].T - dx[:r] @ iPx2.T                                       # :470
    grad3 = Pg2 * dg[r:] - dx[r:] * iPx2                                           # :471
    max_abs_grad = max(np.max(np.abs(grad1)), _max0(np.abs(grad2)), _max0(np.abs(grad3)))   # :472-474
    step0 = step / (max_abs_grad + tiny)                                           # :475
    newU1 = U1 - U1 @ (step0 * grad1)                                              # :476
    newU2 = U2 - U1 @ (step0 * grad2) - step0 * grad3.T * U2                       # :477
    newu3 = u3 - step0 * grad3 * u3                                                # :478
    return (np.concatenate([newL1, newL2], 0), newl3, np.concatenate([newU1, newU2], 1), newu3)   # :480


def _max0(x):
    return np.max(x) if x.size else x.dtype.type(0)


def precond_grad_splu(L12, l3, U12, u3, grads):
    """psgd.py:483-524."""
    grad = [np.reshape(g, (-1, 1)) for g in grads]                                 # :495
    lens = [g.shape[0] for g in grad]                                              # :496
    grad = np.concatenate(grad, 0).astype(L12.dtype)                               # :497
    r, L1, L2, U1, U2 = _splu_blocks(L12, U12)                                     # :499-503
    Ug1 = U1 @ grad[:r] + U2 @ grad[r:]                                            # :506
    Ug2 = u3 * grad[r:]                                                            # :507
    Qg1 = L1 @ Ug1                                                                 # :509
    Qg2 = L2 @ Ug1 + l3 * Ug2                                                      # :510
    LtQg1 = L1.T @ Qg1 + L2.T @ Qg2                                                # :512
    LtQg2 = l3 * Qg2                                                               # :513
    pre_grad = np.concatenate([U1.T @ LtQg1, U2.T @ LtQg1 + u3 * LtQg2], 0)        # :515-516
    pre_grads, idx = [], 0                                                         # :518-522
    for i in range(len(grads)):
        pre_grads.append(np.reshape(pre_grad[idx:idx + lens[i]], np.shape(grads[i])))
        idx += lens[i]
    return pre_grads


# --------------------------------------------------------------------------- UVd
def IpUVtmatvec(U, V, x):
    """psgd.py:540-544: (I + U V') x."""
    return x + U @ (V.T @ x)


def precond_grad_UVd_math(U, V, d, g):
    """psgd.py:619-627 (steps A1..A5 of SURVEY 2.3)."""
    g = IpUVtmatvec(U, V, d * g)                                                 # :625
    return d * IpUVtmatvec(V, U, g)                                              # :626


def update_precond_UVd_math_(U, V, d, v, h, step, tiny, balance=False, update_U=True):
    """psgd.py:554-617.  Mutates U or V, and d, in place; returns None.

    balance  <-> the branch `tf.random.uniform([]) < 0.01` (:562)
    update_U <-> the branch `tf.random.uniform([]) < 0.5`  (:588)"""
    dt = U.dtype.type
    step = dt(step)
    tiny = dt(tiny)
    if balance:                                                                  # :562-567
        rho = np.sqrt(np.max(np.abs(U)) / np.max(np.abs(V)))
        U[...] = U / rho
        V[...] = rho * V

    Qh = IpUVtmatvec(U, V, d * h)                                                # :569
    Ph = d * IpUVtmatvec(V, U, Qh)                                               # :570

    VtU = V.T @ U                                                                # :574
    IpVtU = np.eye(VtU.shape[0], dtype=VtU.dtype) + VtU                          # :575
    invQtv = v / d                                                               # :576
    # first solve has adjoint=True (:577), the second has not (:578)
    invQtv = invQtv - V @ np.linalg.solve(IpVtU.T, U.T @ invQtv).astype(U.dtype)
    invPv = invQtv - U @ np.linalg.solve(IpVtU, V.T @ invQtv).astype(U.dtype)
    invPv = invPv / d                                                            # :579

    nablaD = Ph * h - v * invPv                                                  # :581
    mu = step / (np.max(np.abs(nablaD)) + tiny)                                  # :582
    a, b = Qh, invQtv                                                            # :587 (computed with the OLD d)
    d -= mu * d * nablaD                                                         # :584

    if update_U:                                                                 # :588-601
        atV = a.T @ V
        atVVt = atV @ V.T
        btV = b.T @ V
        btVVt = btV @ V.T
        norm = np.sqrt(np.abs((a.T @ a) * (atVVt @ atVVt.T)
                              + (b.T @ b) * (btVVt @ btVVt.T)
                              - 2 * (a.T @ b) * (atVVt @ btVVt.T)))
        mu = step / (norm + tiny)
        U -= mu * (a @ (atV @ IpVtU) - b @ (btV @ IpVtU))
    else:                                                                        # :602-615
        atU = a.T @ U
        btU = b.T @ U
        UUta = U @ atU.T
        UUtb = U @ btU.T
        norm = np.sqrt(np.abs((UUta.T @ UUta) * (a.T @ a)
                              + (UUtb.T @ UUtb) * (b.T @ b)
                              - 2 * (UUta.T @ UUtb) * (a.T @ b)))
        mu = step / (norm + tiny)
        V -= mu * ((a + V @ atU.T) @ atU - (b + V @ btU.T) @ btU)
    return None


# --------------------------------------------------------------------------- UVd class: index logic only
def uvd_param_index(shapes):
    """psgd.py:684-686: per-parameter sizes and their cumulative sums, in list order."""
    sizes = [int(np.prod(s, dtype=np.int64)) for s in shapes]
    cumsizes = np.cumsum(np.asarray(sizes, dtype=np.int64))
    return sizes, cumsizes


def uvd_flatten(tensors, dtype):
    """psgd.py:729-730,747: concat(reshape(t, [-1])) in parameter order."""
    return np.concatenate([np.reshape(np.asarray(t, dtype=dtype), (-1,)) for t in tensors], 0)


def uvd_unflatten(flat, shapes):
    """psgd.py:758-759: slice [j-i:j] per parameter, reshape to its shape."""
    sizes, cumsizes = uvd_param_index(shapes)
    return [np.reshape(flat[j - i:j], s) for (s, i, j) in zip(shapes, sizes, cumsizes)]


def uvd_init_scales(num_params, rank, dtype):
    """psgd.py:687: std of the initial U, V entries."""
    return np.dtype(dtype).type((1.0 / (num_params * rank)) ** 0.5)


def uvd_clip_lr(pre_grad, lr_params, max_norm, tiny):
    """psgd.py:750-754."""
    dt = pre_grad.dtype.type
    if np.isinf(max_norm):
        return dt(lr_params)
    grad_norm = np.sqrt(np.sum(pre_grad * pre_grad)) + dt(tiny)
    return dt(lr_params) * min(dt(max_norm) / grad_norm, dt(1.0))


def uvd_step(params, grads, Hvs, vs, U, V, d, lr_params, lr_preconditioner, max_norm, tiny,
             balance=False, update_U=True, update_Q=True, exact=True, delta_param_scale=None):
    """psgd.py:729-762 of UVd.step, from the point where the closure has been evaluated: `grads` (and, when the
    preconditioner is updated, the probe vectors `vs` and Hessian-vector products `Hvs`) are given per parameter.
    Updates U, V, d in place (:732-736) and returns the list of updated parameters (:757-762).  With exact=False the
    parameters passed in are the PERTURBED ones (:720) and the perturbation is removed again (:761-762)."""
    dt = U.dtype
    shapes = [np.shape(p) for p in params]
    if update_Q:
        v = uvd_flatten(vs, dt)[:, None]                                          # :729
        h = uvd_flatten(Hvs, dt)[:, None]                                         # :730
        if not exact:                                                             # :735-736
            v, h = v / dt.type(delta_param_scale), h / dt.type(delta_param_scale)
        update_precond_UVd_math_(U, V, d, v, h, lr_preconditioner, tiny, balance=balance, update_U=update_U)
    grad = uvd_flatten(grads, dt)[:, None]                                        # :747
    pre_grad = precond_grad_UVd_math(U, V, d, grad)                               # :748
    lr = uvd_clip_lr(pre_grad, lr_params, max_norm, tiny)                         # :750-754
    deltas = uvd_unflatten(lr * pre_grad[:, 0], shapes)                           # :758-759
    if exact or not update_Q:
        return [np.asarray(p, dtype=dt) - dl for p, dl in zip(params, deltas)]
    return [np.asarray(p, dtype=dt) - (dl + np.asarray(v_, dtype=dt)) for p, dl, v_ in zip(params, deltas, vs)]   # :761-762
