"""GPU parity at BASELINE.json's full sizes (configs[1]: UVd N = 100M, r = 20; configs[4]: Kron 4096 x 4096).

The NumPy oracle cannot hold these on the host in seconds, so the checks are
  * size-independent properties of the HIP results themselves (P = Q'Q is symmetric positive: <g1, P g2> =
    <P g1, g2>, <g, P g> = |Q g|^2 through the separate IpUVtmatvec entry point; linearity), and
  * an independent fp64 run of the reference op sequence (oracle/psgd_oracle_torch.py: one torch call per TF op,
    here on the GPU in fp64 through rocBLAS -- a different code path from the HIP kernels under test).
Tolerances as everywhere: 1e-5 (fp32 paths), 2e-2 (bf16-operand Kron apply).  Skipped on devices without the
memory for the fp64 copies.
"""
import pytest
import torch

from oracle import psgd_oracle_torch as ref64

pytestmark = pytest.mark.gpu

TINY32 = 1.1754943508222875e-38


def _rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float(torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(b))


def _need_gb(gb):
    free, _ = torch.cuda.mem_get_info()
    if free < gb * 2**30:
        pytest.skip("needs %d GiB of free device memory" % gb)


@pytest.fixture(scope="module")
def psgd(hip_lib):
    import preconditioned_stochastic_gradient_descent as m
    return m


def _uvd_inputs(N, r, dev):
    """bench.py's synthetic inputs after a few updates' worth of structure: non-trivial d."""
    g = torch.Generator(device=dev).manual_seed(0)
    sc = 2.0 * (1.0 / (N * r)) ** 0.5
    U = torch.randn(N, r, device=dev, generator=g) * sc
    V = torch.randn(N, r, device=dev, generator=g) * sc
    d = torch.exp(0.3 * torch.randn(N, 1, device=dev, generator=g))
    gr = torch.randn(N, 1, device=dev, generator=g)
    v = torch.randn(N, 1, device=dev, generator=g)
    h = v * torch.exp(torch.empty(N, 1, device=dev).uniform_(-4.6, 4.6, generator=g))
    return U, V, d, gr, v, h


def test_uvd_full_size_properties(psgd):
    """BASELINE configs[1] (N = 100M, r = 20), no reference involved: size-independent properties of the results."""
    _need_gb(40)
    dev = torch.device("cuda:0")
    N, r = 100_000_000, 20
    U, V, d, g, v, h = _uvd_inputs(N, r, dev)
    g2 = torch.roll(g, 12345, 0) * 0.5 + 0.25
    Pg, Pg2 = psgd.precond_grad_UVd_math(U, V, d, g), psgd.precond_grad_UVd_math(U, V, d, g2)
    dot = lambda a, b: float(torch.sum(a.double() * b.double()))
    assert abs(dot(g, Pg2) - dot(Pg, g2)) <= 1e-5 * (dot(g, Pg) * dot(g2, Pg2)) ** 0.5      # P symmetric
    Qg = psgd.IpUVtmatvec(U, V, d * g)
    assert abs(dot(g, Pg) - dot(Qg, Qg)) <= 1e-5 * dot(Qg, Qg)                              # P = Q'Q, positive
    lin = psgd.precond_grad_UVd_math(U, V, d, 0.75 * g - 1.5 * g2)
    assert _rel(lin, 0.75 * Pg - 1.5 * Pg2) < 1e-5                                           # linearity
    assert torch.equal(psgd.precond_grad_UVd_math(U, V, d, g), Pg)                           # run-to-run bitwise
    del g2, Pg2, Qg, lin

    # update: only one factor and d move (psgd.py:586), v and h are read-only, and the fused update -> apply call
    # (what bench.py times) agrees with the two separate calls
    for update_U in (True, False):
        U0, V0, d0, v0, h0 = U.clone(), V.clone(), d.clone(), v.clone(), h.clone()
        U1, V1, d1 = U.clone(), V.clone(), d.clone()
        psgd.update_precond_UVd_math_(U, V, d, v, h, 0.01, TINY32, balance=False, update_U=update_U)
        assert torch.equal(V, V0) if update_U else torch.equal(U, U0)
        assert not torch.equal(U, U0) if update_U else not torch.equal(V, V0)
        assert torch.equal(v, v0) and torch.equal(h, h0)
        assert float(torch.max(torch.abs(d / d0 - 1.0))) <= 0.01 * (1 + 1e-5)                # |mu d nablaD| <= step d (:582-584)
        out = psgd.precond_grad_UVd_math(U, V, d, g)
        out_f = psgd.update_precond_UVd_math_and_precond_grad(U1, V1, d1, v, h, g, 0.01, TINY32, balance=False,
                                                              update_U=update_U)
        assert _rel(out_f, out) < 2e-6
        for name, a_, b_ in (("U", U1, U), ("V", V1, V), ("d", d1, d)):
            assert _rel(a_, b_) < 1e-6, (name, update_U)
        del U0, V0, d0, v0, h0, U1, V1, d1


@pytest.mark.parametrize("N,r", [(10_000_000, 20), (1_000_000, 10)])      # C2, and C4's rank at a tenth of its rows
def test_uvd_large_matches_fp64_restatement(psgd, N, r):
    """Independent fp64 run of psgd.py:619-627 and :554-617 (torch ops on the GPU) on the same inputs."""
    _need_gb(24 if N > 1_000_000 else 4)
    dev = torch.device("cuda:0")
    U, V, d, g, v, h = _uvd_inputs(N, r, dev)
    U64, V64, d64 = U.double(), V.double(), d.double()
    assert _rel(psgd.precond_grad_UVd_math(U, V, d, g), ref64.precond_grad_UVd_math(U64, V64, d64, g.double())) < 1e-5
    for update_U in (True, False):
        psgd.update_precond_UVd_math_(U, V, d, v, h, 0.01, TINY32, balance=False, update_U=update_U)
        ref64.update_precond_UVd_math_(U64, V64, d64, v.double(), h.double(), 0.01, TINY32, balance=False,
                                       update_U=update_U)
        for name, a, b in (("U", U, U64), ("V", V, V64), ("d", d, d64)):
            assert _rel(a, b) < 1e-5, (name, update_U)
    # the fused update -> apply call, balance branch included
    out = psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, g, 0.01, TINY32, balance=True, update_U=True)
    ref64.update_precond_UVd_math_(U64, V64, d64, v.double(), h.double(), 0.01, TINY32, balance=True, update_U=True)
    assert _rel(out, ref64.precond_grad_UVd_math(U64, V64, d64, g.double())) < 1e-5
    for name, a, b in (("U", U, U64), ("V", V, V64), ("d", d, d64)):
        assert _rel(a, b) < 2e-5, name                                    # three chained fp32 updates


def _uvd_inputs_strong_lowrank(N, r, dev, c):
    """Inputs on which the low-rank term is what the tolerance measures: ||U V'||_2 ~ c^2 = O(1) (entries ~ c / sqrt(N),
    sqrt(r) times psgd.py:687's init), V correlated with U so that K = I + V'U (psgd.py:575) is far from the identity,
    and gradients with an O(1) component in range(V) / range(U).  With reference-init scales the low-rank term
    contributes only ~ gain^2 / sqrt(N r) of the output, which a 1e-5 tolerance barely sees at large N."""
    g = torch.Generator(device=dev).manual_seed(11)
    U = torch.randn(N, r, device=dev, generator=g) * (c / N ** 0.5)
    R = torch.linalg.qr(torch.randn(r, r, device=dev, generator=g))[0]
    V = 0.6 * (U @ R) + torch.randn(N, r, device=dev, generator=g) * (0.8 * c / N ** 0.5)
    d = torch.exp(0.3 * torch.randn(N, 1, device=dev, generator=g))
    a = torch.randn(r, 1, device=dev, generator=g)
    gr = 0.5 * torch.randn(N, 1, device=dev, generator=g) + (V @ a) * (N ** 0.5 / c) / d      # O(1) entries, half in range(V)
    v = torch.randn(N, 1, device=dev, generator=g)
    h = v * torch.exp(torch.empty(N, 1, device=dev).uniform_(-2.3, 2.3, generator=g)) + (U @ a) * (N ** 0.5 / c)
    return U.contiguous(), V.contiguous(), d, gr.contiguous(), v, h.contiguous()


@pytest.mark.parametrize("N,r,c", [(1_000_000, 10, 1.0), (4_000_000, 20, 1.3), (1_000_003, 32, 0.8)])
def test_uvd_strong_lowrank_matches_fp64_restatement(psgd, N, r, c):
    """||U V'|| = O(1) at N >= 1e6: apply, both update branches and the fused step against the fp64 restatement."""
    _need_gb(16)
    dev = torch.device("cuda:0")
    U, V, d, g, v, h = _uvd_inputs_strong_lowrank(N, r, dev, c)
    U64, V64, d64, g64 = U.double(), V.double(), d.double(), g.double()
    t64 = d64 * g64
    low = float(torch.linalg.vector_norm(U64 @ (V64.t() @ t64)) / torch.linalg.vector_norm(t64))
    assert low > 0.2, low                               # the low-rank term is a large part of Q (d .* g)
    KmI = V64.t() @ U64
    assert float(torch.linalg.matrix_norm(KmI, 2)) > 0.3
    want = ref64.precond_grad_UVd_math(U64, V64, d64, g64)
    assert _rel(psgd.precond_grad_UVd_math(U, V, d, g), want) < 1e-5
    for update_U in (True, False):
        old32 = {"U": U.double(), "V": V.double(), "d": d.double()}
        old64 = {"U": U64.clone(), "V": V64.clone(), "d": d64.clone()}
        psgd.update_precond_UVd_math_(U, V, d, v, h, 0.01, TINY32, balance=False, update_U=update_U)
        ref64.update_precond_UVd_math_(U64, V64, d64, v.double(), h.double(), 0.01, TINY32, balance=False,
                                       update_U=update_U)
        for name, a_, b_ in (("U", U, U64), ("V", V, V64), ("d", d, d64)):
            assert _rel(a_, b_) < 1e-5, (name, update_U)
            if name != ("V" if update_U else "U"):      # the increment itself (~ step of the state): 2e-3 as in test_uvd_gpu
                assert _rel(a_.double() - old32[name], b_ - old64[name]) < 2e-3, (name, update_U)
    out = psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, g, 0.01, TINY32, balance=False, update_U=True)
    ref64.update_precond_UVd_math_(U64, V64, d64, v.double(), h.double(), 0.01, TINY32, balance=False, update_U=True)
    assert _rel(out, ref64.precond_grad_UVd_math(U64, V64, d64, g64)) < 1e-5


def _tri(n, dev, gen, off):
    return torch.triu(torch.randn(n, n, device=dev, generator=gen) * off, 1) + \
        torch.diag(torch.exp(0.3 * torch.randn(n, device=dev, generator=gen)))


def test_kron_4096_full_size(psgd):
    """BASELINE configs[4] (Transformer-scale 4096 x 4096 factor pair): fp32 update and apply, bf16-operand apply and update."""
    _need_gb(12)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(4096)
    M = N = 4096
    Ql, Qr = _tri(M, dev, gen, 0.01), _tri(N, dev, gen, 0.01)
    dX = torch.randn(M, N, device=dev, generator=gen)
    dG = torch.exp(torch.empty(M, 1, device=dev).uniform_(-1, 1, generator=gen)) * dX * \
        torch.exp(torch.empty(1, N, device=dev).uniform_(-1, 1, generator=gen))
    G = torch.randn(M, N, device=dev, generator=gen)

    want = ref64.precond_grad_dense_dense(Ql.double(), Qr.double(), G.double())
    assert _rel(psgd.precond_grad_kron(Ql, Qr, G), want) < 1e-5
    Gb = G.to(torch.bfloat16)
    want_b = ref64.precond_grad_dense_dense(Ql.double(), Qr.double(), Gb.double())
    out_b = psgd.precond_grad_kron(Ql, Qr, Gb)
    assert out_b.dtype == torch.bfloat16 and _rel(out_b, want_b) < 2e-2

    a, b = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
    a64, b64 = ref64.update_precond_dense_dense(Ql.double(), Qr.double(), dX.double(), dG.double(), 0.01, TINY32)
    assert _rel(a, a64) < 1e-5 and _rel(b, b64) < 1e-5
    assert float(torch.max(torch.abs(torch.tril(a, -1)))) == 0.0 and float(torch.max(torch.abs(torch.tril(b, -1)))) == 0.0
    # and the preconditioned gradient with the updated factors (the mnist_with_lenet5.py:51-53 call pattern)
    assert _rel(psgd.precond_grad_kron(a, b, G), ref64.precond_grad_dense_dense(a64, b64, G.double())) < 1e-5

    # bf16-operand update on the same (bf16-rounded) data: factors within step x bf16 of the fp64 update, increments
    # within the 2e-2 bf16 bar, still upper triangular
    dXb, dGb = dX.to(torch.bfloat16), dG.to(torch.bfloat16)
    ab, bb = psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01)
    a64b, b64b = ref64.update_precond_dense_dense(Ql.double(), Qr.double(), dXb.double(), dGb.double(), 0.01, TINY32)
    assert ab.dtype == torch.float32 and _rel(ab, a64b) < 2e-4 and _rel(bb, b64b) < 2e-4
    rho = (torch.diagonal(Ql).max() / torch.diagonal(Qr).max()).double().sqrt()
    assert _rel(ab.double() - Ql.double() / rho, a64b - Ql.double() / rho) < 2e-2
    assert _rel(bb.double() - Qr.double() * rho, b64b - Qr.double() * rho) < 2e-2
    assert float(torch.max(torch.abs(torch.tril(ab, -1)))) == 0.0 and float(torch.max(torch.abs(torch.tril(bb, -1)))) == 0.0


def test_splu_large(psgd):
    """Sparse LU at a size where every sweep runs many tiles per wave (N = 20M, r = 10): properties + fp64 dense-free
    restatement (the block formulas of psgd.py:396-524 written with torch ops in fp64)."""
    _need_gb(8)
    dev = torch.device("cuda:0")
    N, r = 20_000_003, 10
    gen = torch.Generator(device=dev).manual_seed(7)
    sc = 0.3 / r ** 0.5
    L12 = torch.randn(N, r, device=dev, generator=gen) * (sc * 3 * (r / N) ** 0.5)
    U12 = torch.randn(r, N, device=dev, generator=gen) * (sc * 3 * (r / N) ** 0.5)
    L12[:r] = torch.tril(torch.randn(r, r, device=dev, generator=gen) * sc, -1) + torch.eye(r, device=dev)
    U12[:, :r] = torch.triu(torch.randn(r, r, device=dev, generator=gen) * sc, 1) + torch.eye(r, device=dev)
    l3 = torch.exp(torch.empty(N - r, 1, device=dev).uniform_(-0.5, 0.5, generator=gen))
    u3 = torch.exp(torch.empty(N - r, 1, device=dev).uniform_(-0.5, 0.5, generator=gen)) * 0.7
    g = torch.randn(N, 1, device=dev, generator=gen)
    g2 = torch.randn(N, 1, device=dev, generator=gen)

    def apply64(L12, l3, U12, u3, x):                      # psgd.py:505-516
        L1, L2, U1, U2 = L12[:r], L12[r:], U12[:, :r], U12[:, r:]
        Ug1 = U1 @ x[:r] + U2 @ x[r:]
        Qg1 = L1 @ Ug1
        Qg2 = L2 @ Ug1 + l3 * (u3 * x[r:])
        Lt1 = L1.t() @ Qg1 + L2.t() @ Qg2
        return torch.cat([U1.t() @ Lt1, U2.t() @ Lt1 + u3 * (l3 * Qg2)], 0)

    Pg = psgd.precond_grad_splu(L12, l3, U12, u3, [g])[0]
    Pg2 = psgd.precond_grad_splu(L12, l3, U12, u3, [g2])[0]
    dot = lambda a, b: float(torch.sum(a.double() * b.double()))
    assert abs(dot(g, Pg2) - dot(Pg, g2)) <= 1e-5 * (dot(g, Pg) * dot(g2, Pg2)) ** 0.5      # P symmetric
    assert dot(g, Pg) > 0
    assert _rel(Pg, apply64(L12.double(), l3.double(), U12.double(), u3.double(), g.double())) < 1e-5
    # update: the new factors still define a symmetric positive P and stay close to the old ones (step 0.01)
    new = psgd.update_precond_splu(L12, l3, U12, u3, [g2], [g2 * torch.exp(torch.sin(g))], 0.01)
    Pn, Pn2 = psgd.precond_grad_splu(*new, [g])[0], psgd.precond_grad_splu(*new, [g2])[0]
    assert abs(dot(g, Pn2) - dot(Pn, g2)) <= 1e-5 * (dot(g, Pn) * dot(g2, Pn2)) ** 0.5
    assert _rel(Pn, apply64(*[t.double() for t in new], g.double())) < 1e-5
    assert 0 < _rel(Pn, Pg) < 0.2


@pytest.mark.parametrize("r", [64])
def test_wide_rank_at_36M_rows_against_fp64(psgd, r):
    """Ranks 33 .. 64 (round 5: whole-matrix kernels, the one-sweep Gram, psgd_uvd_wide_update_f32) at a size where element offsets
    pass 2^31 (36 M x 64 = 2.3e9 floats per factor): update (both branches) and apply against the fp64 run of the reference's op
    sequence on the GPU."""
    _need_gb(100)
    dev = torch.device("cuda:0")
    N = 36_000_000
    U, V, d, g, v, h = _uvd_inputs(N, r, dev)
    for upd in (True, False):
        U64, V64, d64 = U.double(), V.double(), d.double()
        psgd.update_precond_UVd_math_(U, V, d, v, h, 0.01, TINY32, balance=False, update_U=upd)
        ref64.update_precond_UVd_math_(U64, V64, d64, v.double(), h.double(), 0.01, TINY32, balance=False, update_U=upd)
        assert max(_rel(U, U64), _rel(V, V64), _rel(d, d64)) < 2e-5, upd
        del U64, V64, d64
    out = psgd.precond_grad_UVd_math(U, V, d, g)
    assert _rel(out, ref64.precond_grad_UVd_math(U.double(), V.double(), d.double(), g.double())) < 1e-5
    # the last rows (the partial tile of a ragged N is covered elsewhere; here: the last whole tiles of a 64-bit offset)
    assert torch.isfinite(U[-64:]).all() and torch.isfinite(out[-64:]).all()
