"""GPU parity: HIP sparse-LU path (through the C ABI) vs the CPU oracle on identical seeded inputs
(update_precond_splu / precond_grad_splu, psgd.py:396-524).

Tolerance (BASELINE.json north_star): preconditioned gradient within 1e-5 relative (norm-wise) of the
fp64 oracle; the updated factors within 1e-5 as well, and the *increment* of an update (a quantity
~step of the state) within 2e-3 of the fp64 increment.
"""
import numpy as np
import pytest
import torch

from oracle import psgd_oracle as orc
from tests.splu_cases import make_splu_problem
from tests.uvd_cases import rel_err

pytestmark = pytest.mark.gpu

APPLY_TOL = 1e-5
STATE_TOL = 1e-5
INCR_TOL = 2e-3

# (N, r): no tail (N = r), tails shorter than the alignment head, odd ranks (scalar head path), tile
# edges, every load-width class, r = 1, r = 32 and r = 64
SHAPES = [(12, 12), (13, 12), (14, 11), (23, 1), (40, 5), (64, 10), (1000, 3), (5000, 7), (4099, 9), (10007, 10),
          (20000, 17), (3000, 32), (50000, 31), (100003, 20), (300001, 10),
          # ranks above 32 (the reference has no limit, psgd.py:420); column chunks of L2 and U2' (splu_wide.py): no tail, one
          # chunk padded, three chunks
          # round 5: ranks 33 .. 64 run on the native tail kernels (64-row tiles), above 64 on the chunks
          (40, 40), (41, 40), (44, 40), (5000, 33), (20011, 48), (3001, 70), (6007, 64), (64, 64), (65, 64), (100003, 40),
          (50021, 64), (30011, 57), (70000, 63), (4001, 100), (9257, 41), (30751, 47), (273, 41)]
KEYS = ("L12", "l3", "U12", "u3")


def _dev(p):
    return {k: torch.from_numpy(v).cuda() for k, v in p.items()}


def _f64(p):
    return {k: v.astype(np.float64) for k, v in p.items()}


@pytest.fixture(scope="module")
def psgd(hip_lib):
    import preconditioned_stochastic_gradient_descent as m
    assert torch.cuda.is_available()
    return m


@pytest.mark.parametrize("N,r", SHAPES)
def test_precond_grad_matches_oracle(psgd, N, r):
    p = make_splu_problem(N, r, seed=N + r)
    t, q = _dev(p), _f64(p)
    out = psgd.precond_grad_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["g"]])
    assert len(out) == 1 and out[0].shape == t["g"].shape and out[0].dtype == torch.float32
    ref = orc.precond_grad_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["g"]])[0]
    assert rel_err(out[0].cpu().numpy(), ref) < APPLY_TOL
    for k in KEYS + ("g",):
        assert np.array_equal(t[k].cpu().numpy(), p[k])


@pytest.mark.parametrize("N,r", SHAPES)
@pytest.mark.parametrize("step", [0.01, 0.1])
def test_update_matches_oracle(psgd, N, r, step):
    p = make_splu_problem(N, r, seed=3 * N + r)
    t, q = _dev(p), _f64(p)
    new = psgd.update_precond_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["dx"]], [t["dg"]], step)
    ref = orc.update_precond_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["dx"]], [q["dg"]], step)
    rho = np.sqrt(max(np.max(np.diag(q["L12"][:r])), np.max(q["l3"], initial=-np.inf)) /
                  max(np.max(np.diag(q["U12"][:, :r])), np.max(q["u3"], initial=-np.inf)))
    base = {"L12": q["L12"] / rho, "l3": q["l3"] / rho, "U12": q["U12"] * rho, "u3": q["u3"] * rho}
    for k, a, b in zip(KEYS, new, ref):
        assert a.shape == t[k].shape and a.dtype == torch.float32
        got = a.cpu().numpy()
        if b.size == 0:
            continue
        assert rel_err(got, b) < STATE_TOL, k
        assert rel_err(got - base[k], b - base[k]) < INCR_TOL, k
    L1, U1 = new[0][:r].cpu().numpy(), new[2][:, :r].cpu().numpy()
    assert np.array_equal(np.triu(L1, 1), np.zeros((r, r))) and np.array_equal(np.tril(U1, -1), np.zeros((r, r)))
    for k in KEYS + ("dx", "dg"):                                  # pure (psgd.py:480 returns new tensors)
        assert np.array_equal(t[k].cpu().numpy(), p[k])


@pytest.mark.parametrize("N,r", [(5000, 33), (20011, 48), (6007, 64)])
def test_chunk_route_of_ranks_33_to_64_agrees_with_the_native_kernels(psgd, monkeypatch, N, r):
    """Ranks 33 .. 64 run on the native tail kernels since round 5; PSGD_SPLU_CHUNKS=1 keeps the column-chunk route of splu_wide.py
    (what ranks above 64 take) reachable for them: both against the oracle and against each other."""
    p = make_splu_problem(N, r, seed=7 * N + r)
    t, q = _dev(p), _f64(p)
    ref_a = orc.precond_grad_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["g"]])[0]
    ref_u = orc.update_precond_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["dx"]], [q["dg"]], 0.01)
    got = {}
    for route in ("native", "chunks"):
        monkeypatch.delenv("PSGD_SPLU_CHUNKS", raising=False)
        if route == "chunks":
            monkeypatch.setenv("PSGD_SPLU_CHUNKS", "1")
        a = psgd.precond_grad_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["g"]])[0].cpu().numpy()
        u = [x.cpu().numpy() for x in psgd.update_precond_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["dx"]], [t["dg"]], 0.01)]
        assert rel_err(a, ref_a) < APPLY_TOL, route
        for k, x, b in zip(KEYS, u, ref_u):
            assert rel_err(x, b) < STATE_TOL, (route, k)
        got[route] = [a] + u
    for x, y in zip(got["native"], got["chunks"]):
        assert rel_err(x, y) < STATE_TOL


@pytest.mark.parametrize("r", list(range(1, 65)))
def test_every_rank_with_a_partial_last_tile(psgd, r):
    """Regression (round 5, found by the randomised sweep): at r = 41 and r = 47 the update left the rows of the partial last tile
    unwritten -- the wave that owns that tile lost its lane index (uvd_kernels.h: wave_in_block).  Every native rank, a tail of 1 ..
    63 rows behind two whole tiles, and a second shape where the owner wave has no whole tile of its own."""
    for N in (r + 64 * 2 + 1 + (r * 7) % 63 + ((32 - r % 32) % 32), 257 + r):
        p = make_splu_problem(N, r, seed=11 * N + r)
        t, q = _dev(p), _f64(p)
        new = psgd.update_precond_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["dx"]], [t["dg"]], 0.05)
        ref = orc.update_precond_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["dx"]], [q["dg"]], 0.05)
        for k, a, b in zip(KEYS, new, ref):
            if b.size:
                assert rel_err(a.cpu().numpy(), b) < STATE_TOL, (N, k)
        out = psgd.precond_grad_splu(*new, [t["g"]])[0]
        assert rel_err(out.cpu().numpy(), orc.precond_grad_splu(*ref, [q["g"]])[0]) < 2 * APPLY_TOL, N


def test_demo_initial_state(psgd):
    """demo_usage_of_all_preconditioners.py:47-51 state, tensor-decomposition parameter count (R*(I+J+K) = 400)."""
    N, r = 400, 10
    p = make_splu_problem(N, r, seed=9, init_like_demo=True)
    t, q = _dev(p), _f64(p)
    new = psgd.update_precond_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["dx"]], [t["dg"]], 0.1)
    ref = orc.update_precond_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["dx"]], [q["dg"]], 0.1)
    for k, a, b in zip(KEYS, new, ref):
        assert rel_err(a.cpu().numpy(), b) < STATE_TOL, k


def test_list_plumbing_restores_shapes(psgd):
    """psgd.py:426-427 and :495-497,:518-522 with the demo's three factor matrices (R x I, R x J, R x K)."""
    shapes = [(5, 10), (5, 20), (5, 50)]
    N, r = 400, 10
    p = make_splu_problem(N, r, seed=4)
    t, q = _dev(p), _f64(p)
    cuts = np.cumsum([0] + [a * b for a, b in shapes])
    split = lambda v: [v[cuts[i]:cuts[i + 1]].reshape(shapes[i]) for i in range(3)]
    outs = psgd.precond_grad_splu(t["L12"], t["l3"], t["U12"], t["u3"], split(t["g"]))
    refs = orc.precond_grad_splu(q["L12"], q["l3"], q["U12"], q["u3"], split(q["g"]))
    assert [tuple(o.shape) for o in outs] == shapes
    for a, b in zip(outs, refs):
        assert rel_err(a.cpu().numpy(), b) < APPLY_TOL
    new = psgd.update_precond_splu(t["L12"], t["l3"], t["U12"], t["u3"], split(t["dx"]), split(t["dg"]), 0.05)
    ref = orc.update_precond_splu(q["L12"], q["l3"], q["U12"], q["u3"], split(q["dx"]), split(q["dg"]), 0.05)
    for a, b in zip(new, ref):
        assert rel_err(a.cpu().numpy(), b) < STATE_TOL


def test_update_apply_sequence(psgd):
    """Ten update -> apply rounds (the demo's opt_step call pattern, demo_usage_of_all_preconditioners.py:61-62)
    against the fp64 oracle run on the same input stream."""
    N, r = 20000, 10
    p = make_splu_problem(N, r, seed=6, init_like_demo=True)
    t, q = _dev(p), _f64(p)
    st = [t[k] for k in KEYS]
    sq = [q[k] for k in KEYS]
    rng = np.random.default_rng(77)
    for _ in range(10):
        dx = rng.standard_normal((N, 1)).astype(np.float32)
        dg = (np.exp(rng.uniform(np.log(0.1), np.log(10.0), (N, 1))) * dx).astype(np.float32)
        st = psgd.update_precond_splu(*st, [torch.from_numpy(dx).cuda()], [torch.from_numpy(dg).cuda()], 0.1)
        sq = orc.update_precond_splu(*sq, [dx.astype(np.float64)], [dg.astype(np.float64)], 0.1)
    for k, a, b in zip(KEYS, st, sq):
        assert rel_err(a.cpu().numpy(), b) < 5e-5, k
    out = psgd.precond_grad_splu(*st, [t["g"]])[0]
    assert rel_err(out.cpu().numpy(), orc.precond_grad_splu(*sq, [q["g"]])[0]) < 5e-5


def test_run_to_run_bitwise_reproducible(psgd):
    p = make_splu_problem(100003, 20, seed=1)
    t = _dev(p)
    a = psgd.precond_grad_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["g"]])[0]
    b = psgd.precond_grad_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["g"]])[0]
    assert torch.equal(a, b)
    u1 = psgd.update_precond_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["dx"]], [t["dg"]], 0.1)
    u2 = psgd.update_precond_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["dx"]], [t["dg"]], 0.1)
    assert all(torch.equal(x, y) for x, y in zip(u1, u2))


def test_in_place_update_through_the_c_abi(psgd):
    """include/psgd_hip.h: the *_new buffers may alias the inputs."""
    from psgd_tf_amd import _lib
    N, r = 5003, 7
    p = make_splu_problem(N, r, seed=8)
    t = _dev(p)
    want = psgd.update_precond_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["dx"]], [t["dg"]], 0.1)
    lib = _lib.load()
    ws = torch.empty(int(lib.psgd_splu_workspace_bytes(N, r)), dtype=torch.uint8, device="cuda")
    a = [t[k].clone() for k in KEYS]
    ptr = [x.data_ptr() for x in a]
    rc = lib.psgd_splu_update_f32(*ptr, t["dx"].data_ptr(), t["dg"].data_ptr(), *ptr, N, r, 0.1, float(psgd._tiny),
                                  ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    for x, y in zip(a, want):
        assert torch.equal(x, y)


def test_rejects_bad_arguments(psgd):
    from psgd_tf_amd._lib import PsgdHipError
    t = _dev(make_splu_problem(100, 4))
    with pytest.raises(PsgdHipError):
        psgd.precond_grad_splu(t["L12"].cpu(), t["l3"].cpu(), t["U12"].cpu(), t["u3"].cpu(), [t["g"].cpu()])
    with pytest.raises(TypeError):
        psgd.precond_grad_splu(t["L12"].double(), t["l3"].double(), t["U12"].double(), t["u3"].double(), [t["g"].double()])
    with pytest.raises(ValueError):
        psgd.precond_grad_splu(t["L12"], t["l3"][:50], t["U12"], t["u3"], [t["g"]])            # l3 length
    with pytest.raises(ValueError):
        psgd.precond_grad_splu(t["L12"], t["l3"], t["U12"], t["u3"], [t["g"][:50]])           # list too short
    with pytest.raises(ValueError):
        psgd.precond_grad_splu(t["L12"], t["l3"], t["U12"][:, :50], t["u3"], [t["g"]])        # U12 shape
    # rank > 32 is NOT refused any more: it runs on column chunks (splu_wide.py; covered by the shape lists above)
    big = _dev(make_splu_problem(100, 33))
    assert psgd.precond_grad_splu(big["L12"], big["l3"], big["U12"], big["u3"], [big["g"]])[0].shape == big["g"].shape
