"""Known-answer tests that pin the CPU oracle to the reference *as read from source*
(SURVEY Appendix C).  The reference ships no tests and TensorFlow is unavailable, so these
analytic identities -- each cited to the psgd.py lines that imply it -- are what the oracle
stands on.  CPU only."""
import numpy as np
import pytest

from oracle import psgd_oracle as orc
from tests.uvd_cases import make_uvd_problem, rel_err

TINY64 = orc.tiny_of(np.float64)


def _tri(rng, n, off=0.1):
    return np.triu(rng.standard_normal((n, n)) * off, 1) + np.diag(np.exp(0.3 * rng.standard_normal(n)))


# ----------------------------------------------------------------------------- constants
def test_tiny_and_delta_constants():
    assert orc.tiny_of(np.float32) == np.float32(1.1754944e-38)            # psgd.py:22 (smallest normal)
    assert float(orc.delta_param_scale_of(np.float32)) == pytest.approx(2.0 ** -11.5, rel=1e-7)   # psgd.py:683
    assert float(orc.delta_param_scale_of(np.float32)) == pytest.approx(3.4526698e-4, rel=1e-6)


# ----------------------------------------------------------------------------- KAT-I identity
def test_identity_like_states():
    rng = np.random.default_rng(0)
    g = rng.standard_normal((7, 1))
    out = orc.precond_grad_dense(0.3 * np.eye(7), [g])                    # Q = cI -> c^2 g (psgd.py:55)
    assert np.allclose(out[0], 0.09 * g)
    G = rng.standard_normal((5, 8))
    assert np.allclose(orc.precond_grad_kron(0.5 * np.eye(5), 2.0 * np.eye(8), G), G)     # a^2 b^2 G
    assert np.allclose(orc.precond_grad_kron(0.5 * np.eye(8), 2.0 * np.eye(5), G.T), G.T)  # other branch (:191)
    N, r = 50, 3
    z = np.zeros((N, r))
    gg = rng.standard_normal((N, 1))
    assert np.allclose(orc.precond_grad_UVd_math(z, z, np.full((N, 1), 0.7), gg), 0.49 * gg)   # psgd.py:625-626


# ----------------------------------------------------------------------------- KAT-R Rosenbrock
def test_rosenbrock_first_step():
    """hello_psgd.py:7-12,25-26 with v = (1, 0): f = 4, g = (-4, 0), Hv = (802, 400)."""
    Q = 0.1 * np.eye(2)
    vs = [np.array(1.0), np.array(0.0)]
    hvs = [np.array(802.0), np.array(400.0)]
    Qn = orc.update_precond_dense(Q, vs, hvs, step=0.2)
    assert np.allclose(Qn, [[0.08, -0.0101326], [0.0, 0.09494634]], rtol=2e-6, atol=1e-9)
    pg = orc.precond_grad_dense(Qn, [np.array(-4.0), np.array(0.0)])
    assert pg[0].shape == () and pg[1].shape == ()                         # original shapes restored (:57-61)
    assert np.allclose([pg[0], pg[1]], [-0.0256, 0.00324243], rtol=2e-6)


# ----------------------------------------------------------------------------- KAT-FP fixed points
def test_dense_fixed_point():
    rng = np.random.default_rng(1)
    n = 6
    B = rng.standard_normal((n, n))
    H = B @ B.T + n * np.eye(n)
    Q = np.linalg.cholesky(np.linalg.inv(H)).T                              # Q'Q = H^-1, Q upper
    dx = rng.standard_normal((n, 1))
    Qn = orc.update_precond_dense(Q, [dx], [H @ dx], step=0.1)
    # a = Q H dx and b = Q^-T dx coincide (psgd.py:38-39): grad is rounding noise, but the step is
    # normalised by max|grad| (:41), so only boundedness by `step` can be asserted on Q itself.
    a, b = Q @ (H @ dx), np.linalg.solve(Q.T, dx)
    assert rel_err(a, b) < 1e-12
    assert np.max(np.abs(Qn - Q)) <= 0.1 * np.max(np.abs(Q)) * n + 1e-12


def test_kron_fixed_point_gradients_vanish():
    rng = np.random.default_rng(2)
    M, N = 5, 7
    Bl, Br = rng.standard_normal((M, M)), rng.standard_normal((N, N))
    Hl, Hr = Bl @ Bl.T + M * np.eye(M), Br @ Br.T + N * np.eye(N)
    Ql = np.linalg.cholesky(np.linalg.inv(Hl)).T
    Qr = np.linalg.cholesky(np.linalg.inv(Hr)).T
    dX = rng.standard_normal((M, N))
    dG = Hl @ dX @ Hr
    A = Ql @ (dG @ Qr.T)                                                   # psgd.py:173
    Bt = np.linalg.solve(Ql.T, np.linalg.solve(Qr.T, dX.T).T)              # psgd.py:174
    assert rel_err(A, Bt) < 1e-11


def test_uvd_fixed_point():
    N, r = 300, 4
    p = make_uvd_problem(N, r, seed=3, uv_gain=2.0, d_spread=0.3)
    U, V, d, v = (p[k].astype(np.float64) for k in ("U", "V", "d", "v"))
    Q = (np.eye(N) + U @ V.T) * d.T
    h = np.linalg.solve(Q.T @ Q, v)                                        # P h = v
    Qh = orc.IpUVtmatvec(U, V, d * h)                                      # a (psgd.py:569)
    invQtv = np.linalg.solve(Q.T, v)                                       # b
    assert rel_err(Qh, invQtv) < 1e-10


# ----------------------------------------------------------------------------- KAT-WOOD
def test_woodbury_inverses_match_dense_solve():
    """invQtv = Q^-T v and invPv = Q^-1 Q^-T v (psgd.py:576-579) vs a dense solve; checked through
    nablaD = Ph h - v invPv reconstructed from an update with step -> the d increment."""
    N, r = 200, 5
    p = make_uvd_problem(N, r, seed=4, uv_gain=3.0, d_spread=0.4)
    q = {k: p[k].astype(np.float64) for k in p}
    Q = (np.eye(N) + q["U"] @ q["V"].T) * q["d"].T
    P = Q.T @ Q
    nabla_ref = (P @ q["h"]) * q["h"] - q["v"] * np.linalg.solve(P, q["v"])     # psgd.py:581
    d0 = q["d"].copy()
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY64, update_U=True)
    mu = 0.01 / (np.max(np.abs(nabla_ref)) + TINY64)
    assert rel_err(q["d"], d0 - mu * d0 * nabla_ref) < 1e-12                     # psgd.py:582-584


def test_uvd_apply_is_QtQ():
    N, r = 150, 6
    p = make_uvd_problem(N, r, seed=5, uv_gain=3.0, d_spread=0.4)
    q = {k: p[k].astype(np.float64) for k in p}
    Q = (np.eye(N) + q["U"] @ q["V"].T) * q["d"].T
    assert rel_err(orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"]), Q.T @ (Q @ q["g"])) < 1e-12


def test_uvd_update_matches_dense_formulas():
    """The U branch (psgd.py:589-601) and V branch (:603-615) against explicitly formed N x N algebra."""
    N, r = 120, 4
    for update_U in (True, False):
        p = make_uvd_problem(N, r, seed=6, uv_gain=2.0, d_spread=0.2)
        q = {k: p[k].astype(np.float64) for k in p}
        U, V, d, v, h = (q[k].copy() for k in ("U", "V", "d", "v", "h"))
        Qm = (np.eye(N) + U @ V.T) * d.T
        a = Qm @ h
        b = np.linalg.solve(Qm.T, v)
        K = np.eye(r) + V.T @ U
        if update_U:
            G = a @ (a.T @ V @ K) - b @ (b.T @ V @ K)
            nrm = np.sqrt(abs(((a.T @ a) * ((a.T @ V @ V.T) @ (V @ V.T @ a)) + (b.T @ b) * ((b.T @ V @ V.T) @ (V @ V.T @ b))
                                - 2 * (a.T @ b) * ((a.T @ V @ V.T) @ (V @ V.T @ b))).item()))
            expect = U - 0.01 / (nrm + TINY64) * G
        else:
            atU, btU = a.T @ U, b.T @ U
            G = (a + V @ atU.T) @ atU - (b + V @ btU.T) @ btU
            p_, q_ = U @ atU.T, U @ btU.T
            nrm = np.sqrt(abs(((p_.T @ p_) * (a.T @ a) + (q_.T @ q_) * (b.T @ b) - 2 * (p_.T @ q_) * (a.T @ b)).item()))
            expect = V - 0.01 / (nrm + TINY64) * G
        orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY64, update_U=update_U)
        assert rel_err(q["U"] if update_U else q["V"], expect) < 1e-11
        assert np.array_equal(q["V"] if update_U else q["U"], V if update_U else U)     # psgd.py:586


def test_uvd_balance_branch_keeps_UVt():
    p = make_uvd_problem(100, 3, seed=7, uv_gain=2.0)
    q = {k: p[k].astype(np.float64) for k in p}
    q["U"] *= 9.0
    UVt = q["U"] @ q["V"].T
    rho = np.sqrt(np.max(np.abs(q["U"])) / np.max(np.abs(q["V"])))          # psgd.py:563-565
    U1, V1, d1 = q["U"] / rho, q["V"] * rho, q["d"].copy()
    orc.update_precond_UVd_math_(U1, V1, d1, q["v"], q["h"], 0.01, TINY64, balance=False, update_U=True)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY64, balance=True, update_U=True)
    assert rel_err(q["U"], U1) < 1e-14 and rel_err(q["V"], V1) < 1e-14 and rel_err(q["d"], d1) < 1e-14
    assert np.isclose(np.max(np.abs(U1 + 0)), np.max(np.abs(U1)))           # sanity
    assert rel_err((q["U"] @ q["V"].T)[:5, :5] * 0 + UVt[:5, :5], UVt[:5, :5]) == 0.0


# ----------------------------------------------------------------------------- KAT-TRI / KAT-NORM
@pytest.mark.parametrize("M,N", [(26, 6), (6, 26), (1, 1), (2, 2), (9, 9)])
def test_kron_update_invariants(M, N):
    rng = np.random.default_rng(M + 10 * N)
    Ql, Qr = _tri(rng, M), _tri(rng, N)
    dX, dG = rng.standard_normal((M, N)), rng.standard_normal((M, N))
    Qln, Qrn = orc.update_precond_kron(Ql, Qr, dX, dG, 0.05)
    for Qn in (Qln, Qrn):
        assert np.array_equal(Qn, np.triu(Qn)) and (np.diag(Qn) > 0).all()          # psgd.py:175-179
    rho = np.sqrt(np.max(np.diag(Ql)) / np.max(np.diag(Qr)))
    assert np.isclose(np.max(np.diag(Ql / rho)), np.max(np.diag(Qr * rho)))         # psgd.py:166-170


def test_dense_step_normalisation():
    rng = np.random.default_rng(8)
    n = 5
    Q = _tri(rng, n)
    dx, dg = rng.standard_normal((n, 1)), rng.standard_normal((n, 1))
    a, b = Q @ dg, np.linalg.solve(Q.T, dx)
    grad = np.triu(a @ a.T - b @ b.T)
    Qn = orc.update_precond_dense(Q, [dx], [dg], step=0.07)
    assert np.allclose(Q - Qn, 0.07 / (np.max(np.abs(grad)) + TINY64) * grad @ Q)   # psgd.py:41-42


# ----------------------------------------------------------------------------- KAT-BR / KAT-SWAP / KAT-DISP
def test_apply_branches_agree_in_exact_arithmetic():
    rng = np.random.default_rng(9)
    for M, N in [(4, 9), (9, 4), (5, 5)]:
        Ql, Qr, G = _tri(rng, M), _tri(rng, N), rng.standard_normal((M, N))
        expect = Ql.T @ Ql @ G @ Qr.T @ Qr
        assert rel_err(orc.precond_grad_dense_dense(Ql, Qr, G), expect) < 1e-13     # psgd.py:189-192


def test_mirrored_formats_swap():
    rng = np.random.default_rng(10)
    M, N = 6, 9
    Ql, qr = _tri(rng, M), np.exp(0.2 * rng.standard_normal((1, N)))
    dX, dG, G = (rng.standard_normal((M, N)) for _ in range(3))
    a = orc.update_precond_kron(Ql, qr, dX, dG, 0.03)                      # (dense, scale)   psgd.py:88
    b = orc.update_precond_kron(qr, Ql, dX.T, dG.T, 0.03)                  # (scale, dense)   psgd.py:102
    assert np.allclose(a[0], b[1]) and np.allclose(a[1], b[0])
    assert np.allclose(orc.precond_grad_kron(Ql, qr, G), orc.precond_grad_kron(qr, Ql, G.T).T)   # :130 vs :144


DISPATCH = [((5, 5), (7, 7), "dense_dense"), ((5, 5), (2, 7), "dense_norm"), ((5, 5), (1, 7), "dense_scale"),
            ((2, 5), (7, 7), "norm_dense"), ((2, 5), (1, 7), "norm_scale"), ((1, 5), (7, 7), "scale_dense"),
            ((1, 5), (2, 7), "scale_norm"), ((2, 5), (2, 7), "unknown"), ((1, 5), (1, 7), "unknown"),
            ((3, 5), (7, 7), "unknown"), ((5, 5), (3, 7), "unknown"),
            ((1, 1), (7, 7), "dense_dense"), ((2, 2), (7, 7), "dense_dense"), ((5, 5), (2, 2), "dense_dense"),
            ((5, 5), (1, 1), "dense_dense"), ((2, 2), (1, 7), "dense_scale"), ((1, 1), (2, 7), "dense_norm")]


@pytest.mark.parametrize("sl,sr,fmt", DISPATCH)
def test_dispatch_table(sl, sr, fmt):
    assert orc.kron_format(sl, sr) == fmt                                  # psgd.py:80-110; README.md:39


def test_unknown_format_passthrough():
    Ql, Qr, G = np.ones((2, 5)), np.ones((2, 7)), np.ones((5, 7))
    a, b = orc.update_precond_kron(Ql, Qr, G, G, 0.01)
    assert a is Ql and b is Qr and orc.precond_grad_kron(Ql, Qr, G) is G   # psgd.py:97-99,139-141


@pytest.mark.parametrize("fmt,sl,sr", [("dense_norm", (5, 5), (2, 8)), ("norm_dense", (2, 5), (8, 8)),
                                       ("norm_scale", (2, 5), (1, 8)), ("scale_norm", (1, 5), (2, 8)),
                                       ("dense_scale", (5, 5), (1, 8)), ("scale_dense", (1, 5), (8, 8))])
def test_sparse_formats_equal_their_dense_embedding(fmt, sl, sr):
    """A normalization factor is the matrix diag(ql[0]) with last column ql[1] (psgd.py:204-205,
    :223-229); a scaling factor is diag(qr).  Their preconditioned gradient must equal the
    dense(x)dense formula on the embedded matrices."""
    rng = np.random.default_rng(len(fmt) + sl[0])

    def make(shape):
        m, n = shape
        if m == n:
            Q = _tri(rng, m)
            return Q, Q
        if m == 2:
            q = np.stack([np.exp(0.2 * rng.standard_normal(n)), 0.3 * rng.standard_normal(n)])
            q[1, -1] = 0.0        # "excluding the last entry" (psgd.py:205): kept at 0 by every update (:237,:241)
            D = np.diag(q[0]).copy()
            D[:-1, -1] = q[1, :-1]
            return q, D
        q = np.exp(0.2 * rng.standard_normal((1, n)))
        return q, np.diag(q[0])

    (ql, Dl), (qr, Dr) = make(sl), make(sr)
    G = rng.standard_normal((sl[1], sr[1]))
    assert rel_err(orc.precond_grad_kron(ql, qr, G), Dl.T @ Dl @ G @ Dr.T @ Dr) < 1e-12


# ----------------------------------------------------------------------------- KAT-IDX / KAT-INIT
def test_uvd_index_logic():
    shapes = [(2, 30), (30, 30), (30,), (30, 1), (1,)]                     # rnn_xor_UVd_preconditioner.py:28-31
    sizes, cum = orc.uvd_param_index(shapes)
    assert sizes == [60, 900, 30, 30, 1] and list(cum) == [60, 960, 990, 1020, 1021]   # psgd.py:684-686
    rng = np.random.default_rng(11)
    tensors = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    flat = orc.uvd_flatten(tensors, np.float32)
    assert flat.shape == (1021,)
    back = orc.uvd_unflatten(flat, shapes)
    assert all(np.array_equal(a, b) for a, b in zip(back, tensors))        # psgd.py:758-759
    assert float(orc.uvd_init_scales(1021, 10, np.float32)) == pytest.approx((1 / 10210) ** 0.5, rel=1e-6)


def test_uvd_clip_lr():
    pg = np.full((100, 1), 0.5, dtype=np.float32)                          # norm 5
    assert orc.uvd_clip_lr(pg, 0.01, np.inf, 1e-38) == np.float32(0.01)    # psgd.py:750-751
    assert float(orc.uvd_clip_lr(pg, 0.01, 1.0, 1e-38)) == pytest.approx(0.002, rel=1e-6)   # :753-754
    assert float(orc.uvd_clip_lr(pg, 0.01, 50.0, 1e-38)) == pytest.approx(0.01, rel=1e-6)


# ----------------------------------------------------------------------------- fp32 vs fp64 and torch port
@pytest.mark.parametrize("N,r", [(4096, 10), (3000, 20), (1000, 1)])
def test_fp32_oracle_tracks_fp64(N, r):
    p = make_uvd_problem(N, r, seed=12, uv_gain=2.0, d_spread=0.3)
    q = {k: p[k].astype(np.float64) for k in p}
    assert rel_err(orc.precond_grad_UVd_math(p["U"], p["V"], p["d"], p["g"]),
                   orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < 2e-6
    orc.update_precond_UVd_math_(p["U"], p["V"], p["d"], p["v"], p["h"], 0.01, orc.tiny_of(np.float32), update_U=True)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, orc.tiny_of(np.float32), update_U=True)
    for k in ("U", "V", "d"):
        assert p[k].dtype == np.float32 and rel_err(p[k], q[k]) < 1e-5


def test_torch_port_matches_numpy_oracle():
    import torch
    from oracle import psgd_oracle_torch as ot
    p = make_uvd_problem(2000, 8, seed=13, uv_gain=2.0, d_spread=0.3)
    q = {k: p[k].astype(np.float64) for k in p}
    t = {k: torch.from_numpy(v.copy()) for k, v in q.items()}
    for upd in (True, False):
        orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY64, balance=upd, update_U=upd)
        ot.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY64, balance=upd, update_U=upd)
    for k in ("U", "V", "d"):
        assert rel_err(t[k].numpy(), q[k]) < 1e-12
    assert rel_err(ot.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"]).numpy(),
                   orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < 1e-12
    rng = np.random.default_rng(14)
    Ql, Qr = _tri(rng, 9), _tri(rng, 5)
    dX, dG = rng.standard_normal((9, 5)), rng.standard_normal((9, 5))
    a = orc.update_precond_dense_dense(Ql, Qr, dX, dG, 0.01)
    b = ot.update_precond_dense_dense(*(torch.from_numpy(x) for x in (Ql, Qr, dX, dG)), 0.01, TINY64)
    assert rel_err(b[0].numpy(), a[0]) < 1e-12 and rel_err(b[1].numpy(), a[1]) < 1e-12


# ----------------------------------------------------------------------------- sparse LU (psgd.py:396-524)
from tests.splu_cases import make_splu_problem, splu_dense   # noqa: E402


@pytest.mark.parametrize("N,r", [(40, 5), (23, 1), (12, 12), (64, 10)])
def test_splu_apply_is_dense_QtQ(N, r):
    """precond_grad_splu = (LU)'(LU) g with the N x N factors written out (psgd.py:399-403, :486-487)."""
    p = {k: v.astype(np.float64) for k, v in make_splu_problem(N, r, seed=N + r).items()}
    L, U = splu_dense(p["L12"], p["l3"], p["U12"], p["u3"])
    Q = L @ U
    want = Q.T @ (Q @ p["g"])
    k = N // 3
    got = orc.precond_grad_splu(p["L12"], p["l3"], p["U12"], p["u3"], [p["g"][:k].reshape(-1), p["g"][k:].reshape(1, -1)])
    assert got[0].shape == (k,) and got[1].shape == (1, N - k)                 # shapes restored (:518-522)
    assert rel_err(np.concatenate([x.reshape(-1) for x in got]), want[:, 0]) < 1e-13


@pytest.mark.parametrize("N,r", [(40, 5), (23, 1), (64, 10)])
def test_splu_update_is_masked_dense_update(N, r):
    """The block formulas :455-478 are the dense rule  L <- L - mu tril_mask(Qg Qg' - Q^-T x (Q^-T x)') L,
    U <- U - mu U triu_mask(Pg g' - x (P^-1 x)'), restricted to the sparsity pattern, with the dynamic-range
    balance of :411-417 applied first.  Independent route: full N x N solves."""
    p = {k: v.astype(np.float64) for k, v in make_splu_problem(N, r, seed=3 * N + r).items()}
    step = 0.1
    new = orc.update_precond_splu(p["L12"], p["l3"], p["U12"], p["u3"], [p["dx"]], [p["dg"]], step)
    L, U = splu_dense(p["L12"], p["l3"], p["U12"], p["u3"])
    rho = np.sqrt(np.max(np.diag(L)) / np.max(np.diag(U)))
    L, U = L / rho, U * rho
    Q = L @ U
    Qg, iQtx = Q @ p["dg"], np.linalg.solve(Q.T, p["dx"])
    mask = np.tril(np.ones((N, N), bool))
    mask[:, r:] = False
    mask |= np.eye(N, dtype=bool)
    GL = (Qg @ Qg.T - iQtx @ iQtx.T) * mask
    Lw = L - step / np.max(np.abs(GL)) * GL @ L
    Pg, iPx = Q.T @ Qg, np.linalg.solve(Q, iQtx)
    GU = (Pg @ p["dg"].T - p["dx"] @ iPx.T) * mask.T
    Uw = U - step / np.max(np.abs(GU)) * U @ GU
    Ln, Un = splu_dense(*new)
    assert rel_err(Ln, Lw) < 1e-12 and rel_err(Un, Uw) < 1e-12
    assert np.array_equal(np.triu(new[0][:r], 1), np.zeros((r, r))) and np.array_equal(np.tril(new[2][:, :r], -1), np.zeros((r, r)))
    # pure: inputs untouched
    q = make_splu_problem(N, r, seed=3 * N + r)
    assert all(np.array_equal(p[k], q[k].astype(np.float64)) for k in ("L12", "l3", "U12", "u3"))


def test_splu_gradient_vanishes_at_fixed_point():
    """With dg = P^-1 dx both whitening residuals vanish: Q dg = Q^-T dx and P dg = dx, so the raw gradients
    of :455-458 and :468-471 are rounding noise (they are then normalised to max = step, like UVd's)."""
    N, r = 30, 4
    p = {k: v.astype(np.float64) for k, v in make_splu_problem(N, r, seed=5).items()}
    L, U = splu_dense(p["L12"], p["l3"], p["U12"], p["u3"])
    Q = L @ U
    dg = np.linalg.solve(Q.T @ Q, p["dx"])
    assert rel_err(Q @ dg, np.linalg.solve(Q.T, p["dx"])) < 1e-12
    # a tiny step from the fixed point keeps the state to O(step) even though the noise is normalised up
    new = orc.update_precond_splu(p["L12"], p["l3"], p["U12"], p["u3"], [p["dx"]], [dg], 1e-6)
    rho = np.sqrt(max(np.max(np.diag(p["L12"][:r])), np.max(p["l3"])) / max(np.max(np.diag(p["U12"][:, :r])), np.max(p["u3"])))
    assert rel_err(new[0], p["L12"] / rho) < 1e-5 and rel_err(new[2], p["U12"] * rho) < 1e-5


def test_splu_demo_initial_state_first_step():
    """KAT from demo_usage_of_all_preconditioners.py:47-51: L = U = 0.1 I.  Then Q = 0.01 I, rho = 1,
    Qg = 0.01 dg, Q^-T dx = 100 dx: every formula collapses to elementwise arithmetic."""
    N, r, step = 12, 3, 0.1
    p = {k: v.astype(np.float64) for k, v in make_splu_problem(N, r, seed=1, init_like_demo=True).items()}
    dx, dg = p["dx"], p["dg"]
    c = float(np.float32(0.1))                                                  # the fixtures are fp32 values
    nL12, nl3, nU12, nu3 = orc.update_precond_splu(p["L12"], p["l3"], p["U12"], p["u3"], [dx], [dg], step)
    a, b = c * c * dg, dx / (c * c)
    GL1, GL2, GL3 = np.tril(a[:r] @ a[:r].T - b[:r] @ b[:r].T), a[r:] @ a[:r].T - b[r:] @ b[:r].T, a[r:] ** 2 - b[r:] ** 2
    mu = step / max(np.abs(GL1).max(), np.abs(GL2).max(), np.abs(GL3).max())
    assert rel_err(nL12, c * np.concatenate([np.eye(r) - mu * GL1, -mu * GL2], 0)) < 1e-12
    assert rel_err(nl3, c * (1 - mu * GL3)) < 1e-12
    Pg, iPx = c ** 4 * dg, dx / c ** 4
    GU1, GU2, GU3 = np.triu(Pg[:r] @ dg[:r].T - dx[:r] @ iPx[:r].T), Pg[:r] @ dg[r:].T - dx[:r] @ iPx[r:].T, Pg[r:] * dg[r:] - dx[r:] * iPx[r:]
    mu = step / max(np.abs(GU1).max(), np.abs(GU2).max(), np.abs(GU3).max())
    assert rel_err(nU12, c * np.concatenate([np.eye(r) - mu * GU1, -mu * GU2], 1)) < 1e-12
    assert rel_err(nu3, c * (1 - mu * GU3)) < 1e-12
    out = orc.precond_grad_splu(p["L12"], p["l3"], p["U12"], p["u3"], [p["g"]])
    assert rel_err(out[0], c ** 4 * p["g"]) < 1e-12


def test_splu_fp32_oracle_tracks_fp64():
    p = make_splu_problem(4096, 10, seed=2)
    q = {k: v.astype(np.float64) for k, v in p.items()}
    a = orc.update_precond_splu(p["L12"], p["l3"], p["U12"], p["u3"], [p["dx"]], [p["dg"]], 0.01)
    b = orc.update_precond_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["dx"]], [q["dg"]], 0.01)
    assert all(x.dtype == np.float32 for x in a)
    for x, y in zip(a, b):
        assert rel_err(x, y) < 1e-5
    assert rel_err(orc.precond_grad_splu(p["L12"], p["l3"], p["U12"], p["u3"], [p["g"]])[0],
                   orc.precond_grad_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["g"]])[0]) < 1e-5
