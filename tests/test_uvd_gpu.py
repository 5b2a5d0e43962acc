"""GPU parity: HIP UVd path (through the C ABI) vs the CPU oracle on identical seeded inputs.

Tolerance (BASELINE.json north_star): preconditioned gradient within 1e-5 relative (norm-wise)
of the fp64 oracle; the updated state U, V, d within 1e-5 as well, and the *increment* of an
update (a quantity ~step = 1e-2 of the state) within 2e-3 of the fp64 increment.
"""
import numpy as np
import pytest
import torch

from oracle import psgd_oracle as orc
from tests.uvd_cases import TINY32, make_uvd_problem, rel_err

pytestmark = pytest.mark.gpu

APPLY_TOL = 1e-5
STATE_TOL = 1e-5
INCR_TOL = 2e-3

# (N, r): tile edges (63/64/65 rows), ragged tails, every load-width class of the tile
# config (r % 4 == 0, r % 2 == 0, odd), the extremes r = 1 and r = 32, N = 1.
SHAPES = [(1, 4), (63, 20), (64, 20), (65, 20), (777, 3), (1000, 1), (1021, 10), (4096, 10), (5000, 20),
          (2049, 32), (1537, 17), (3000, 18), (2500, 9), (4099, 7), (100003, 20), (300001, 10), (50000, 31)]


def _to_dev(p):
    return {k: torch.from_numpy(v).cuda() for k, v in p.items()}


def _f64(p):
    return {k: v.astype(np.float64) for k, v in p.items()}


@pytest.fixture(scope="module")
def psgd(hip_lib):
    import preconditioned_stochastic_gradient_descent as m
    assert torch.cuda.is_available()
    return m


@pytest.mark.parametrize("N,r", SHAPES)
@pytest.mark.parametrize("uv_gain,d_spread", [(1.0, 0.0), (3.0, 0.5)])
def test_precond_grad_matches_oracle(psgd, N, r, uv_gain, d_spread):
    p = make_uvd_problem(N, r, seed=N + r, uv_gain=uv_gain, d_spread=d_spread)
    t = _to_dev(p)
    out = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
    assert out.shape == t["g"].shape and out.dtype == torch.float32
    q = _f64(p)
    ref = orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])
    assert rel_err(out.cpu().numpy(), ref) < APPLY_TOL
    # inputs untouched
    for k in ("U", "V", "d", "g"):
        assert np.array_equal(t[k].cpu().numpy(), p[k])


@pytest.mark.parametrize("N,r", [(1021, 10), (5000, 20), (777, 3), (2049, 32), (300001, 10), (64, 1)])
def test_ipuvt_matvec(psgd, N, r):
    p = make_uvd_problem(N, r, seed=7, uv_gain=3.0 * r ** 0.5)
    t = _to_dev(p)
    out = psgd.IpUVtmatvec(t["U"], t["V"], t["g"])
    q = _f64(p)
    assert rel_err(out.cpu().numpy(), orc.IpUVtmatvec(q["U"], q["V"], q["g"])) < APPLY_TOL
    # a matrix x (psgd.py:542): k columns in one sweep of V and one of U per group of four columns
    cols = [q["g"], q["v"], q["h"] * 1e-2, q["g"] - q["v"], q["d"], q["g"] * q["v"]]
    for k in (2, 4, 5, 6):
        xk64 = np.concatenate(cols[:k], 1)
        xk = torch.from_numpy(xk64.astype(np.float32)).cuda()
        outk = psgd.IpUVtmatvec(t["U"], t["V"], xk)
        assert outk.shape == (N, k) and outk.is_contiguous()
        refk = orc.IpUVtmatvec(q["U"], q["V"], xk.cpu().numpy().astype(np.float64))
        assert rel_err(outk.cpu().numpy(), refk) < APPLY_TOL, k
        for j in range(k):                                   # every column on its own, not only the norm of the block
            assert rel_err(outk[:, j].cpu().numpy(), refk[:, j]) < APPLY_TOL, (k, j)
    # a transposed view as input (non-contiguous columns) is fine too
    xv = torch.from_numpy(np.concatenate(cols[:4], 1).astype(np.float32).T.copy()).cuda().t()
    assert rel_err(psgd.IpUVtmatvec(t["U"], t["V"], xv).cpu().numpy(),
                   orc.IpUVtmatvec(q["U"], q["V"], xv.cpu().numpy().astype(np.float64))) < APPLY_TOL


@pytest.mark.parametrize("N,r", [(1021, 10), (5000, 20), (777, 3), (2049, 32), (300001, 10), (64, 1), (65, 17), (1, 4),
                                 (3001, 40), (2500, 64), (1900, 50)])
def test_precond_grad_with_a_matrix_g(psgd, N, r):
    """psgd.py:619-627 with g a matrix (docstring :623 "either matrices or column vectors"): d broadcasts over the columns
    (:625-626).  Every column against the fp64 oracle at 1e-5, and against the column-vector call on that column (the
    four-column sweeps reduce on the matrix core in another order, so that comparison is to rounding, not bitwise)."""
    p = make_uvd_problem(N, r, seed=5 * N + r, uv_gain=2.0, d_spread=0.4)
    t, q = _to_dev(p), _f64(p)
    cols = [q["g"], q["v"], q["h"] * 1e-2, q["g"] - q["v"], q["d"], q["g"] * q["v"]]
    for k in (2, 4, 5):
        G64 = np.concatenate(cols[:k], 1)
        G = torch.from_numpy(G64.astype(np.float32)).cuda()
        G0 = G.clone()
        out = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], G)
        assert out.shape == (N, k) and out.dtype == torch.float32 and out.is_contiguous()
        assert torch.equal(G, G0)                                                # the input is not modified
        ref = orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], G.cpu().numpy().astype(np.float64))
        assert ref.shape == (N, k)
        for j in range(k):
            assert rel_err(out[:, j].cpu().numpy(), ref[:, j]) < APPLY_TOL, (k, j)
            # ... and the column-vector call on the same column agrees to rounding
            one = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], G[:, j:j + 1].contiguous())
            assert rel_err(out[:, j:j + 1].cpu().numpy(), one.cpu().numpy().astype(np.float64)) < 1e-5, (k, j)
    # a transposed view ([k, N] storage) goes in without a copy; d as [N] instead of [N, 1]
    Gv = torch.from_numpy(np.concatenate(cols[:4], 1).astype(np.float32).T.copy()).cuda().t()
    assert N == 1 or not Gv.is_contiguous()
    out = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"].reshape(-1), Gv)
    ref = orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], Gv.cpu().numpy().astype(np.float64))
    assert rel_err(out.cpu().numpy(), ref) < APPLY_TOL
    with pytest.raises(ValueError):
        psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], torch.zeros(N + 1, 3, device="cuda"))


@pytest.mark.parametrize("N,r", SHAPES)
@pytest.mark.parametrize("update_U", [True, False])
def test_update_matches_oracle(psgd, N, r, update_U):
    # (N < r: K = I + V'U is the identity plus a matrix of rank N; a smaller gain keeps it well conditioned -- psgd.py:574-578
    #  treats such a shape like any other, and so does this test)
    p = make_uvd_problem(N, r, seed=3 * N + r, uv_gain=2.0 if N >= r else 0.5, d_spread=0.3)
    t = _to_dev(p)
    ret = psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY32,
                                        balance=False, update_U=update_U)
    assert ret is None
    q = _f64(p)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False,
                                 update_U=update_U)
    got = {k: t[k].cpu().numpy() for k in ("U", "V", "d")}
    for k in ("U", "V", "d"):
        assert rel_err(got[k], q[k]) < STATE_TOL, k
    changed, frozen = ("U", "V") if update_U else ("V", "U")
    assert np.array_equal(got[frozen], p[frozen])            # only one factor changes (psgd.py:586)
    for k in (changed, "d"):
        inc_ref = q[k] - p[k].astype(np.float64)
        inc_got = got[k].astype(np.float64) - p[k].astype(np.float64)
        assert rel_err(inc_got, inc_ref) < INCR_TOL, k
    # v, h read-only
    assert np.array_equal(t["v"].cpu().numpy(), p["v"]) and np.array_equal(t["h"].cpu().numpy(), p["h"])


@pytest.mark.parametrize("N,r", [(1, 4), (3, 10), (17, 20), (2, 32), (31, 32)])
@pytest.mark.parametrize("update_U", [True, False])
@pytest.mark.parametrize("balance", [False, True])
def test_update_with_fewer_rows_than_rank(psgd, N, r, update_U, balance):
    """N < r (psgd.py:574-578: the r x r solves with K = I + V'U do not care that U, V have fewer rows than columns):
    the update, then the apply and the fused update -> apply on the updated state, against the fp64 oracle."""
    p = make_uvd_problem(N, r, seed=7 * N + r, uv_gain=0.5, d_spread=0.3)
    if balance:
        p["U"] *= 3.0
    t, q = _to_dev(p), _f64(p)
    psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY32, balance=balance, update_U=update_U)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=balance, update_U=update_U)
    for k in ("U", "V", "d"):
        assert rel_err(t[k].cpu().numpy(), q[k]) < STATE_TOL, k
    out = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
    assert rel_err(out.cpu().numpy(), orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < APPLY_TOL
    outf = psgd.update_precond_UVd_math_and_precond_grad(t["U"], t["V"], t["d"], t["v"], t["h"], t["g"], 0.01, TINY32,
                                                         balance=False, update_U=not update_U)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=not update_U)
    for k in ("U", "V", "d"):
        assert rel_err(t[k].cpu().numpy(), q[k]) < STATE_TOL, k
    assert rel_err(outf.cpu().numpy(), orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < APPLY_TOL


@pytest.mark.parametrize("N,r", [(5000, 20), (1021, 10), (4099, 7)])
def test_update_balance_branch(psgd, N, r):
    p = make_uvd_problem(N, r, seed=11, uv_gain=2.0)
    p["U"] *= 7.0       # make rho != 1
    t = _to_dev(p)
    psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY32, balance=True, update_U=True)
    q = _f64(p)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=True, update_U=True)
    for k in ("U", "V", "d"):
        assert rel_err(t[k].cpu().numpy(), q[k]) < STATE_TOL, k


def test_update_then_apply_sequence(psgd):
    """Ten alternating updates followed by an apply, against the fp64 oracle run on the same stream
    of inputs (the UVd.step call pattern, psgd.py:732 -> :748)."""
    N, r = 20000, 10
    p = make_uvd_problem(N, r, seed=5)
    t = _to_dev(p)
    q = _f64(p)
    rng = np.random.default_rng(99)
    for it in range(10):
        v = rng.standard_normal((N, 1)).astype(np.float32)
        h = (np.exp(rng.uniform(np.log(1e-2), np.log(1e2), (N, 1))) * v).astype(np.float32)
        upd = (it % 2 == 0)
        psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], torch.from_numpy(v).cuda(),
                                      torch.from_numpy(h).cuda(), 0.01, TINY32, balance=(it == 4), update_U=upd)
        orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], v.astype(np.float64), h.astype(np.float64), 0.01,
                                     TINY32, balance=(it == 4), update_U=upd)
    for k in ("U", "V", "d"):
        assert rel_err(t[k].cpu().numpy(), q[k]) < 5e-5, k
    out = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
    ref = orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])
    assert rel_err(out.cpu().numpy(), ref) < 5e-5


def test_run_to_run_bitwise_reproducible(psgd):
    p = make_uvd_problem(100003, 20, seed=1)
    t = _to_dev(p)
    a = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
    b = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
    assert torch.equal(a, b)


def test_fixed_point_leaves_state_unchanged(psgd):
    """KAT-FP: with h = P^-1 v the update gradient vanishes (a = b, nablaD = 0)."""
    N, r = 3000, 5
    p = make_uvd_problem(N, r, seed=2, uv_gain=2.0, d_spread=0.2)
    q = _f64(p)
    Q = (np.eye(N) + q["U"] @ q["V"].T) * q["d"].T                   # Q = (I + U V') diag(d)
    h = np.linalg.solve(Q.T @ Q, q["v"])                             # P h = v
    t = _to_dev(p)
    t["h"] = torch.from_numpy(h.astype(np.float32)).cuda()
    psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY32, balance=False,
                                  update_U=True)
    # nablaD is pure rounding noise, normalised to max |.| = step: d moves by at most step, U by ~noise
    assert rel_err(t["d"].cpu().numpy(), p["d"]) < 0.011
    assert np.isfinite(t["U"].cpu().numpy()).all()


def test_rejects_bad_arguments(psgd):
    from psgd_tf_amd._lib import PsgdHipError
    p = make_uvd_problem(100, 4)
    t = _to_dev(p)
    with pytest.raises(PsgdHipError):
        psgd.precond_grad_UVd_math(t["U"].cpu(), t["V"].cpu(), t["d"].cpu(), t["g"].cpu())
    with pytest.raises(TypeError):
        psgd.precond_grad_UVd_math(t["U"].double(), t["V"].double(), t["d"].double(), t["g"].double())
    # a row-slice that starts at an odd row of an r = 3 matrix is not 16-byte aligned: refused, not mis-read
    p3 = _to_dev(make_uvd_problem(101, 3))
    with pytest.raises(PsgdHipError, match="16-byte"):
        psgd.precond_grad_UVd_math(p3["U"][1:], p3["V"][1:], p3["d"][1:], p3["g"][1:])
    with pytest.raises(ValueError):
        psgd.precond_grad_UVd_math(t["U"], t["V"][:50], t["d"], t["g"])               # shape mismatch


@pytest.mark.parametrize("N,r", [(3001, 10), (2048, 40)])
def test_strided_views_are_accepted(psgd, N, r):
    """The reference's ops take whatever tensor they are given; here strided views (every second column of a wider buffer,
    a column of a matrix) are copied on the way in and the in-place state is written back on the way out: same results as on
    contiguous tensors, and the memory between the view's elements is untouched."""
    p = make_uvd_problem(N, r, seed=9, uv_gain=2.0, d_spread=0.3)
    t = _to_dev(p)
    wide = {k: torch.full((N, 2 * r), 7.0, device="cuda") for k in ("U", "V")}
    for k in ("U", "V"):
        wide[k][:, ::2] = t[k]
    cols = torch.full((N, 4), 7.0, device="cuda")
    for j, k in enumerate(("d", "g", "v", "h")):
        cols[:, j] = t[k][:, 0]
    Uv, Vv = wide["U"][:, ::2], wide["V"][:, ::2]
    dv, gv, vv, hv = (cols[:, j:j + 1] for j in range(4))
    assert not Uv.is_contiguous() and not dv.is_contiguous()
    out_v = psgd.precond_grad_UVd_math(Uv, Vv, dv, gv)
    assert torch.equal(out_v, psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"]))
    assert torch.equal(psgd.IpUVtmatvec(Uv, Vv, gv), psgd.IpUVtmatvec(t["U"], t["V"], t["g"]))
    for upd in (True, False):
        psgd.update_precond_UVd_math_(Uv, Vv, dv, vv, hv, 0.01, TINY32, balance=False, update_U=upd)
        psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY32, balance=False, update_U=upd)
    o1 = psgd.update_precond_UVd_math_and_precond_grad(Uv, Vv, dv, vv, hv, gv, 0.01, TINY32, balance=False, update_U=True)
    o2 = psgd.update_precond_UVd_math_and_precond_grad(t["U"], t["V"], t["d"], t["v"], t["h"], t["g"], 0.01, TINY32,
                                                       balance=False, update_U=True)
    assert torch.equal(o1, o2)
    assert torch.equal(wide["U"][:, ::2], t["U"]) and torch.equal(wide["V"][:, ::2], t["V"]) and torch.equal(cols[:, 0:1], t["d"])
    assert bool((wide["U"][:, 1::2] == 7.0).all()) and bool((wide["V"][:, 1::2] == 7.0).all())
    assert torch.equal(cols[:, 1:2], t["g"])                       # read-only operands untouched
    q = {k: v.astype(np.float64) for k, v in p.items()}
    for upd in (True, False, True):
        orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=upd)
    assert rel_err(t["U"].cpu().numpy(), q["U"]) < 1e-5 and rel_err(t["d"].cpu().numpy(), q["d"]) < 1e-5


@pytest.mark.parametrize("N,r", [(1021, 10), (5000, 20), (4099, 7), (100003, 20), (2049, 32), (777, 3)])
@pytest.mark.parametrize("update_U", [True, False])
def test_fused_update_apply_matches_oracle_and_unfused(psgd, N, r, update_U):
    """the fused call = update_precond_UVd_math_ then precond_grad_UVd_math (psgd.py:732 -> :748)."""
    p = make_uvd_problem(N, r, seed=N + 7 * r, uv_gain=2.0, d_spread=0.3)
    a, b = _to_dev(p), _to_dev(p)
    out_f = psgd.update_precond_UVd_math_and_precond_grad(a["U"], a["V"], a["d"], a["v"], a["h"], a["g"], 0.01, TINY32,
                                                          balance=False, update_U=update_U)
    psgd.update_precond_UVd_math_(b["U"], b["V"], b["d"], b["v"], b["h"], 0.01, TINY32, balance=False, update_U=update_U)
    out_u = psgd.precond_grad_UVd_math(b["U"], b["V"], b["d"], b["g"])
    for k in ("U", "V", "d"):                                   # same arithmetic, separately compiled kernel variant
        assert rel_err(a[k].cpu().numpy(), b[k].cpu().numpy()) < 1e-6, k
    assert rel_err(out_f.cpu().numpy(), out_u.cpu().numpy()) < 2e-6
    q = _f64(p)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=update_U)
    assert rel_err(out_f.cpu().numpy(), orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < APPLY_TOL


@pytest.mark.parametrize("N,r", [(1021, 1), (3000, 5), (4099, 8), (5000, 9), (1021, 10), (4096, 16), (3001, 17), (5000, 20),
                                 (2500, 24), (2600, 25), (2049, 32)])
@pytest.mark.parametrize("update_U", [True, False])
def test_register_row_coef_kernel_equals_block_reference(psgd, hip_lib, N, r, update_U):
    """The r x r algebra of the update (two pivoted LU solves + the norm, psgd.py:574-615) has two kernels: the
    one-row-per-lane register form the entry points launch and the block-cooperative reference form
    (psgd_set_tuning(2, 1)).  Same elimination order element by element; only the inner products are summed in a
    different order (fp64), so the updated states agree far inside fp32 rounding.  K = I + V'U is made far from the
    identity (correlated U, V, gain 3) so that pivoting and both triangular phases matter."""
    p = make_uvd_problem(N, r, seed=5 * N + r, uv_gain=3.0 * r ** 0.5, d_spread=0.3)
    p["V"] = (0.7 * p["U"] @ np.linalg.qr(np.random.default_rng(r).standard_normal((r, r)))[0] + 0.5 * p["V"]).astype(np.float32)
    a, b = _to_dev(p), _to_dev(p)
    psgd.update_precond_UVd_math_(a["U"], a["V"], a["d"], a["v"], a["h"], 0.01, TINY32, balance=False, update_U=update_U)
    try:
        assert hip_lib.psgd_set_tuning(2, 1) == 0
        psgd.update_precond_UVd_math_(b["U"], b["V"], b["d"], b["v"], b["h"], 0.01, TINY32, balance=False, update_U=update_U)
        torch.cuda.synchronize()
    finally:
        hip_lib.psgd_set_tuning(2, 0)
    for k in ("U", "V", "d"):
        assert rel_err(a[k].cpu().numpy(), b[k].cpu().numpy()) < 1e-7, k
    q = _f64(p)
    KmI = q["V"].T @ q["U"]
    assert np.linalg.norm(KmI, 2) > 0.5                                   # K is not a perturbation of I here
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=update_U)
    for k in ("U", "V", "d"):
        assert rel_err(a[k].cpu().numpy(), q[k]) < STATE_TOL, k


def test_strided_building_blocks_of_the_wide_rank_path(hip_lib):
    """The `*_ld` entry points (include/psgd_hip.h): column VIEWS of a wider row-major matrix as operands -- column sums, axpy,
    rank-2 update and the Gram sweep against torch on the same views; what is outside the view is never written; a view that
    is not aligned to its rank's access width is refused (PSGD_ERR_ALIGN), never mis-read."""
    import ctypes
    from psgd_tf_amd import _lib, uvd_wide
    import preconditioned_stochastic_gradient_descent as m
    torch.manual_seed(3)
    for N, rtot, lo, rc in ((5000, 40, 20, 20), (3001, 50, 25, 25), (1000, 36, 18, 18), (777, 96, 32, 32), (64, 33, 22, 11), (130, 40, 0, 20)):
        W = torch.randn(N, rtot, device="cuda") * 0.1
        W0 = W.clone()
        view = W[:, lo:lo + rc]
        cx = uvd_wide._Ctx(torch.empty(N, rc, device="cuda"), m.uvd_workspace)
        assert cx.rc == rc and cx.c == 1
        x, y = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
        S = cx.colsums(view, [x, y])
        want = torch.stack([view.double().t() @ x.double(), view.double().t() @ y.double()])
        assert float((S - want).norm() / want.norm()) < 1e-6
        out = [x.clone(), y.clone()]
        coef = torch.randn(2, rc, device="cuda")
        cx.axpy(view, out, coef)
        for j, base in enumerate((x, y)):
            w_ = base.double() + view.double() @ coef[j].double()
            assert float((out[j].double() - w_).norm() / w_.norm()) < 1e-6
        c1, c2 = torch.randn(rc, device="cuda"), torch.randn(rc, device="cuda")
        want = view.double() - (torch.outer(x.double(), c1.double()) - torch.outer(y.double(), c2.double()))
        cx.rank2(view, x, y, c1, c2)
        assert float((view.double() - want).norm() / want.norm()) < 1e-6
        keep = torch.ones(rtot, dtype=torch.bool, device="cuda")
        keep[lo:lo + rc] = False
        assert torch.equal(W[:, keep], W0[:, keep])                       # nothing outside the view was touched
    # a misaligned view of a rank with 16-byte accesses
    W = torch.randn(512, 41, device="cuda")
    cx = uvd_wide._Ctx(torch.empty(512, 20, device="cuda"), m.uvd_workspace)
    with pytest.raises(_lib.PsgdHipError):
        cx.colsums(W[:, 1:21], [torch.randn(512, device="cuda")])


@pytest.mark.parametrize("N,r", [(5000, 48), (3001, 33), (2049, 64), (1021, 100), (100003, 40), (4001, 70), (2500, 65), (1, 40), (3, 50),
                                 (31, 64), (32, 57), (200003, 64), (70016, 41)])
def test_wide_rank_matches_oracle(psgd, N, r):
    """r > 32 (the reference has no rank limit, psgd.py:663): column chunks of U and V through the same HIP kernels
    (psgd_tf_amd/uvd_wide.py).  Apply, IpUVtmatvec (vector and matrix), both update branches, the balance branch and
    the fused call against the fp64 oracle, with K = I + V'U far from the identity."""
    p = make_uvd_problem(N, r, seed=N + r, uv_gain=2.0 * r ** 0.5, d_spread=0.3)
    p["V"] = (0.6 * p["U"] @ np.linalg.qr(np.random.default_rng(r).standard_normal((r, r)))[0] + 0.6 * p["V"]).astype(np.float32)
    t, q = _to_dev(p), _f64(p)
    assert np.linalg.norm(q["V"].T @ q["U"], 2) > 0.5
    out = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
    assert out.shape == t["g"].shape
    assert rel_err(out.cpu().numpy(), orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < APPLY_TOL
    assert rel_err(psgd.IpUVtmatvec(t["U"], t["V"], t["g"]).cpu().numpy(), orc.IpUVtmatvec(q["U"], q["V"], q["g"])) < APPLY_TOL
    x3 = torch.cat([t["g"], t["v"], t["h"] * 1e-2], 1).contiguous()
    assert rel_err(psgd.IpUVtmatvec(t["U"], t["V"], x3).cpu().numpy(),
                   orc.IpUVtmatvec(q["U"], q["V"], x3.cpu().numpy().astype(np.float64))) < APPLY_TOL
    for bal, upd in ((False, True), (False, False), (True, True)):
        before = {k: t[k].cpu().numpy().astype(np.float64) for k in ("U", "V", "d")}
        before64 = {k: q[k].copy() for k in ("U", "V", "d")}
        assert psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY32, balance=bal,
                                             update_U=upd) is None
        orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=bal, update_U=upd)
        for k in ("U", "V", "d"):
            assert rel_err(t[k].cpu().numpy(), q[k]) < 2 * STATE_TOL, (k, bal, upd)
        if not bal:
            for k in (("U" if upd else "V"), "d"):
                assert rel_err(t[k].cpu().numpy() - before[k], q[k] - before64[k]) < INCR_TOL, (k, upd)
            frozen = "V" if upd else "U"
            assert np.array_equal(t[frozen].cpu().numpy().astype(np.float64), before[frozen])
    outf = psgd.update_precond_UVd_math_and_precond_grad(t["U"], t["V"], t["d"], t["v"], t["h"], t["g"], 0.01, TINY32,
                                                         balance=False, update_U=False)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=False)
    assert rel_err(outf.cpu().numpy(), orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < 2 * APPLY_TOL


def test_uvd_class_wide_rank(hip_lib):
    """class UVd with rank_of_modification = 48 on a convex quadratic: runs, stays finite, loss goes down."""
    import preconditioned_stochastic_gradient_descent as psgd
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    A = torch.randn(300, 300, device=dev) * 0.1
    H = A @ A.t() + 0.3 * torch.eye(300, device=dev)
    w = torch.randn(300, 1, device=dev).requires_grad_(True)
    opt = psgd.UVd([w], rank_of_modification=48, lr_params=0.05, lr_preconditioner=0.05, generator=torch.Generator().manual_seed(2))
    closure = lambda: 0.5 * (w.t() @ H @ w).sum()
    l0 = float(opt.step(closure).detach())
    for _ in range(80):
        l = float(opt.step(closure).detach())
    assert opt._U.shape == (300, 48) and torch.isfinite(opt._U).all() and l < 0.3 * l0


@pytest.mark.parametrize("N,r", [(1, 33), (31, 40), (32, 64), (33, 64), (1000, 33), (4099, 40), (65536, 48), (100003, 57), (300001, 64)])
def test_wide_gram_matches_fp64(hip_lib, N, r):
    """psgd_uvd_gram_wide_f32 (ranks 33 .. 64, one sweep): every inner product of the columns of [U | V | d.*h | v./d] (psgd.py:569-615)
    against the fp64 product of the same fp32 matrices.  The bf16 x 3 split is exact, the products are accumulated in fp32 chains of
    256 rows and folded in fp64: a few 1e-7 of sqrt(G_ii G_jj).  Fewer rows than a tile, exactly one tile, ragged ends, many
    workgroups."""
    from psgd_tf_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(N + r)
    U = torch.randn(N, r, device="cuda", generator=g) * 3e-3
    V = torch.randn(N, r, device="cuda", generator=g) * 2e-1
    V[:, 0] = 0.0                                                   # a zero column stays exactly zero
    d = torch.rand(N, device="cuda", generator=g) + 0.5
    v, h = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    n = int(hip_lib.psgd_uvd_gram_wide_scratch_bytes(N, r))
    assert n > 0
    scr = torch.empty(n, dtype=torch.uint8, device="cuda")
    G = torch.full((2 * r + 2, 2 * r + 2), float("nan"), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(hip_lib.psgd_uvd_gram_wide_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(), h.data_ptr(), N, r, G.data_ptr(),
                                              scr.data_ptr(), n, st), "psgd_uvd_gram_wide_f32")
    W = torch.cat([U.double(), V.double(), (d * h).double()[:, None], (v / d).double()[:, None]], 1)
    ref = W.T @ W
    sc = ref.diagonal().sqrt().clamp_min(1e-30)
    assert torch.equal(G, G.T)
    assert float(((G - ref).abs() / (sc[:, None] * sc[None, :])).max()) < 5e-7
    assert torch.equal(G[r], torch.zeros_like(G[r]))                # (column r of W = V[:, 0])
    G2 = torch.empty_like(G)
    _lib.check(hip_lib.psgd_uvd_gram_wide_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(), h.data_ptr(), N, r, G2.data_ptr(),
                                              scr.data_ptr(), n, st), "psgd_uvd_gram_wide_f32")
    assert torch.equal(G, G2)                                       # fixed reduction order
    assert hip_lib.psgd_uvd_gram_wide_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(), h.data_ptr(), N, 32, G.data_ptr(),
                                          scr.data_ptr(), n, st) == _lib.PSGD_ERR_RANK
    assert hip_lib.psgd_uvd_gram_wide_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(), h.data_ptr(), N, r, G.data_ptr(),
                                          scr.data_ptr(), n - 1, st) == _lib.PSGD_ERR_WORKSPACE


@pytest.mark.parametrize("N,r", [(5000, 48), (100003, 40), (2049, 64), (70001, 33)])
@pytest.mark.parametrize("update_U", [True, False])
def test_wide_update_routes_agree(psgd, monkeypatch, N, r, update_U):
    """The three routes of the rank 33 .. 64 update -- psgd_uvd_wide_update_f32 (four launches), the whole-matrix building blocks
    (PSGD_WIDE_UPDATE=0: what a row-sharded update takes) and the column chunks of rounds 3-4 (PSGD_WIDE_FULL=0) -- against the oracle
    and against each other."""
    p = make_uvd_problem(N, r, seed=N + r, uv_gain=2.0 * r ** 0.5, d_spread=0.3)
    q = _f64(p)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=update_U)
    got = {}
    for route, env in (("native", {}), ("blocks", {"PSGD_WIDE_UPDATE": "0"}), ("chunks", {"PSGD_WIDE_FULL": "0"})):
        for k in ("PSGD_WIDE_UPDATE", "PSGD_WIDE_FULL"):
            monkeypatch.delenv(k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        t = _to_dev(p)
        psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY32, balance=False, update_U=update_U)
        got[route] = {k: t[k].cpu().numpy() for k in ("U", "V", "d")}
        for k in ("U", "V", "d"):
            assert rel_err(got[route][k], q[k]) < 2 * STATE_TOL, (route, k)
            assert rel_err(got[route][k] - p[k], q[k] - p[k].astype(np.float64)) < INCR_TOL, (route, k)
        frozen = "V" if update_U else "U"
        assert np.array_equal(got[route][frozen], p[frozen]), route
    for k in ("U", "V", "d"):
        assert rel_err(got["native"][k], got["blocks"][k]) < STATE_TOL
        assert rel_err(got["native"][k], got["chunks"][k]) < STATE_TOL


def test_wide_update_rejects_bad_arguments(hip_lib):
    from psgd_tf_amd import _lib
    N, r = 1000, 40
    U, V = torch.randn(N, r, device="cuda"), torch.randn(N, r, device="cuda")
    d, v, h = torch.ones(N, device="cuda"), torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    n = int(hip_lib.psgd_uvd_wide_update_scratch_bytes(N, r))
    scr = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    call = lambda Uq, rr, nn: hip_lib.psgd_uvd_wide_update_f32(Uq, V.data_ptr(), d.data_ptr(), v.data_ptr(), h.data_ptr(), N, rr, 0.01, 1e-30,
                                                               1, scr.data_ptr(), nn, st)
    assert hip_lib.psgd_uvd_wide_update_scratch_bytes(N, 32) == _lib.PSGD_ERR_RANK
    assert hip_lib.psgd_uvd_wide_update_scratch_bytes(N, 65) == _lib.PSGD_ERR_RANK
    assert call(U.data_ptr(), 20, n) == _lib.PSGD_ERR_RANK
    assert call(U.data_ptr(), r, n - 1) == _lib.PSGD_ERR_WORKSPACE
    assert call(U.data_ptr() + 4, r, n) == _lib.PSGD_ERR_ALIGN
    assert call(0, r, n) == _lib.PSGD_ERR_BAD_ARG
    assert call(U.data_ptr(), r, n) == _lib.PSGD_OK
    torch.cuda.synchronize()


@pytest.mark.parametrize("r", list(range(1, 65)))
def test_every_rank_with_a_partial_last_tile(psgd, r):
    """Every specialised rank (1 .. 32) and every whole-matrix rank (33 .. 64) with partial last tiles -- one shape where the wave
    that owns the tail has whole tiles of its own, one where it has none (the case that lost its lane index in the sparse-LU update at
    r = 41 / 47, uvd_kernels.h: wave_in_block): apply and both update branches against the oracle."""
    for N in (9000 + 13 * r, 257 + r):
        p = make_uvd_problem(N, r, seed=5 * N + r, uv_gain=1.5, d_spread=0.3)
        for upd in (True, False):
            t, q = _to_dev(p), _f64(p)
            psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY32, balance=False, update_U=upd)
            orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=upd)
            for k in ("U", "V", "d"):
                assert rel_err(t[k].cpu().numpy(), q[k]) < 2 * STATE_TOL, (N, upd, k)
                assert rel_err(t[k].cpu().numpy() - p[k], q[k] - p[k].astype(np.float64)) < INCR_TOL or k == ("V" if upd else "U"), (N, upd, k)
            out = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
            assert rel_err(out.cpu().numpy(), orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < 2 * APPLY_TOL, (N, upd)


@pytest.mark.parametrize("N,r", [(5000, 48), (100003, 40), (2049, 64), (70001, 33), (31, 57), (257, 41)])
@pytest.mark.parametrize("update_U", [True, False])
@pytest.mark.parametrize("balance", [False, True])
def test_wide_fused_step_matches_oracle_and_the_two_calls(psgd, monkeypatch, N, r, update_U, balance):
    """psgd_uvd_wide_update_apply_f32 (ranks 33 .. 64: sweep 2 also reduces the sums of the apply, three reads of U and V) against the
    oracle's update followed by its apply, and against the two separate calls (PSGD_WIDE_STEP=0)."""
    p = make_uvd_problem(N, r, seed=N + 3 * r, uv_gain=2.0 * r ** 0.5, d_spread=0.3)
    q = _f64(p)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=balance, update_U=update_U)
    ref = orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])
    got = {}
    for route in ("fused", "two_calls"):
        monkeypatch.delenv("PSGD_WIDE_STEP", raising=False)
        if route == "two_calls":
            monkeypatch.setenv("PSGD_WIDE_STEP", "0")
        t = _to_dev(p)
        out = psgd.update_precond_UVd_math_and_precond_grad(t["U"], t["V"], t["d"], t["v"], t["h"], t["g"], 0.01, TINY32,
                                                            balance=balance, update_U=update_U)
        assert out.shape == t["g"].shape
        assert rel_err(out.cpu().numpy(), ref) < 2 * APPLY_TOL, route
        for k in ("U", "V", "d"):
            assert rel_err(t[k].cpu().numpy(), q[k]) < 2 * STATE_TOL, (route, k)
        got[route] = [out.cpu().numpy()] + [t[k].cpu().numpy() for k in ("U", "V", "d")]
    for a, b in zip(got["fused"], got["two_calls"]):
        assert rel_err(a, b) < STATE_TOL
