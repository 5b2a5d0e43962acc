"""Error convention of the hot path (reference README.md:56): no exceptions, NaN/Inf propagate silently --
and zero inputs give exact zeros.  Checked for every HIP entry point family."""
import numpy as np
import pytest
import torch

from tests.splu_cases import make_splu_problem
from tests.uvd_cases import TINY32, make_uvd_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def psgd(hip_lib):
    import preconditioned_stochastic_gradient_descent as m
    return m


def _dev(p):
    return {k: torch.from_numpy(v).cuda() for k, v in p.items()}


def test_uvd_nan_propagates_and_zero_stays_zero(psgd):
    t = _dev(make_uvd_problem(5000, 10, seed=1))
    assert float(psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], torch.zeros_like(t["g"])).abs().max()) == 0.0
    g = t["g"].clone()
    g[1234] = float("nan")
    out = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], g)
    assert torch.isnan(out).all()                         # a NaN in g reaches V'(d g), hence every row
    g = t["g"].clone()
    g[77] = float("inf")
    assert not torch.isfinite(psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], g)).all()
    h = t["h"].clone()
    h[5] = float("nan")
    psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], h, 0.01, TINY32, balance=False, update_U=True)
    assert torch.isnan(t["U"]).any() and torch.isnan(t["d"]).any()      # silently poisoned state, no exception


def test_splu_nan_propagates_and_zero_stays_zero(psgd):
    t = _dev(make_splu_problem(5003, 7, seed=2))
    st = [t[k] for k in ("L12", "l3", "U12", "u3")]
    assert float(psgd.precond_grad_splu(*st, [torch.zeros_like(t["g"])])[0].abs().max()) == 0.0
    g = t["g"].clone()
    g[4000] = float("nan")
    assert torch.isnan(psgd.precond_grad_splu(*st, [g])[0]).all()
    dx = t["dx"].clone()
    dx[3] = float("nan")
    new = psgd.update_precond_splu(*st, [dx], [t["dg"]], 0.1)
    assert all(torch.isnan(x).any() for x in new)


def test_kron_nan_propagates_and_zero_stays_zero(psgd):
    rng = np.random.default_rng(0)
    M, N = 151, 16
    Ql = torch.from_numpy((np.triu(rng.standard_normal((M, M)) * 0.05, 1) + np.eye(M)).astype(np.float32)).cuda()
    Qr = torch.from_numpy((np.triu(rng.standard_normal((N, N)) * 0.05, 1) + np.eye(N)).astype(np.float32)).cuda()
    G = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda()
    assert float(psgd.precond_grad_kron(Ql, Qr, torch.zeros_like(G)).abs().max()) == 0.0
    G[7, 3] = float("nan")
    assert torch.isnan(psgd.precond_grad_kron(Ql, Qr, G)).any()
    a, b = psgd.update_precond_kron(Ql, Qr, G, G, 0.01)
    assert torch.isnan(a).any() or torch.isnan(b).any()


@pytest.mark.parametrize("M,N", [(640, 512), (1024, 1536)])
def test_kron_large_and_bf16_paths_propagate_nan(psgd, M, N):
    """The large-problem kernels (bf16 x 3 split GEMMs with the truncating split, register-resident solve strips, the
    bf16-operand apply and update): zero in -> zero out for the apply, a NaN / Inf in the data reaches the outputs."""
    rng = np.random.default_rng(1)
    tri = lambda n: torch.from_numpy((np.triu(rng.standard_normal((n, n)) * (0.3 / n ** 0.5), 1) + np.eye(n)).astype(np.float32)).cuda()
    Ql, Qr = tri(M), tri(N)
    G = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda()
    dX = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda()
    assert float(psgd.precond_grad_kron(Ql, Qr, torch.zeros_like(G)).abs().max()) == 0.0
    assert float(psgd.precond_grad_kron(Ql, Qr, torch.zeros_like(G).bfloat16()).float().abs().max()) == 0.0
    for bad in (float("nan"), float("inf")):
        Gb = G.clone()
        Gb[M // 3, N // 5] = bad
        for data in (Gb, Gb.bfloat16()):
            out = psgd.precond_grad_kron(Ql, Qr, data).float()
            assert not torch.isfinite(out).all()
            a, b = psgd.update_precond_kron(Ql, Qr, dX.to(data.dtype), data, 0.01)
            assert not (torch.isfinite(a).all() and torch.isfinite(b).all())
            a, b = psgd.update_precond_kron(Ql, Qr, data, dX.to(data.dtype), 0.01)      # through the triangular solves
            assert not (torch.isfinite(a).all() and torch.isfinite(b).all())
