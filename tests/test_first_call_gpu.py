"""First call on a poisoned workspace == second call, bit for bit (GPU).

The C ABI never allocates: every call works in a caller-owned workspace, and the launches of a call hand intermediates to one another
through it, some of them across forked streams.  A consumer that runs ahead of its producer, or reads a region no launch of the
call wrote, is invisible on every call but the FIRST on a fresh block (same inputs afterwards: the previous call's intermediates are
in place) -- and torch.empty() often returns zeros, which hides it there too.  Here every workspace the Python boundary creates is
filled with 0xFF bytes (NaN in fp32, f16 and bf16, huge counters), each entry point is called twice on the same inputs, and the two
results must be finite and identical.  (Round 6: a missing event wait in a new stream order of the large Kron update was found this way.)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def psgd():
    import preconditioned_stochastic_gradient_descent as m
    return m


class _Poison:
    """torch.empty on the device returns poisoned blocks while active: 0xFF bytes (uint8 workspaces: NaN in fp32 / f16 / bf16, huge
    counters), NaN (floating outputs and scratch).  reset(): every cached workspace and scratch block is dropped, so the next call of any
    entry point allocates -- and gets poison -- again."""

    def __init__(self, monkeypatch):
        self.made = 0
        orig = torch.empty

        def empty(*a, **k):
            t = orig(*a, **k)
            if t.is_cuda and t.numel():
                if t.dtype == torch.uint8:
                    t.fill_(0xFF)
                    self.made += 1
                elif t.dtype.is_floating_point:
                    t.fill_(float("nan"))
            return t
        monkeypatch.setattr(torch, "empty", empty)
        self.reset()

    def reset(self):
        from psgd_tf_amd import kron, uvd_wide
        from psgd_tf_amd import preconditioned_stochastic_gradient_descent as core
        for cache in (kron._kron_ws, kron._kron_ws_bf16, kron._batch_ws, kron._sparse_ws, core._ws_cache):
            cache._d.clear()
        from psgd_tf_amd import sharded
        for d in (uvd_wide._gram_scratch, uvd_wide._update_scratch, uvd_wide._wide_scratch, kron._prepared, kron._apply_slots,
                  kron._padded_factors, kron._handoff_watch, sharded._backends, sharded._splu_backends):
            d.clear()
        kron.invalidate_factor_cache()
        self.made = 0


@pytest.fixture
def poisoned(monkeypatch):
    p = _Poison(monkeypatch)
    yield p
    monkeypatch.undo()
    p.reset()


def _tri(rng, n, off=0.02, scale=1.0):
    q = np.triu(rng.standard_normal((n, n)) * off, 1) + np.diag(np.exp(0.3 * rng.standard_normal(n)))
    return torch.from_numpy((q * scale).astype(np.float32)).cuda()


def _randn(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32)).cuda()


def _twice(fn, poison, expect_ws=True):
    poison.reset()
    first = fn()
    first = [t.clone() for t in (first if isinstance(first, (tuple, list)) else [first])]
    if expect_ws:
        assert poison.made, "the call made no workspace: nothing was poisoned"
    second = fn()
    second = list(second if isinstance(second, (tuple, list)) else [second])
    torch.cuda.synchronize()
    for a, b in zip(first, second):
        assert torch.isfinite(a.float()).all(), "first call on the poisoned workspace is not finite"
        assert torch.equal(a, b), "first call differs from the second by %g" % float((a.float() - b.float()).abs().max())


KRON_SHAPES = [(512, 512), (600, 530), (1030, 1100), (1024, 2304), (300, 4000), (64, 8192), (2048, 2176), (2560, 2048), (4096, 4096)]


@pytest.mark.parametrize("M,N", KRON_SHAPES)
def test_kron_fp32_apply(psgd, poisoned, M, N):
    rng = np.random.default_rng(M + 7 * N)
    Ql, Qr, G = _tri(rng, M, scale=1.5), _tri(rng, N), _randn(rng, M, N)
    _twice(lambda: psgd.precond_grad_kron(Ql, Qr, G), poisoned)


@pytest.mark.parametrize("M,N", KRON_SHAPES + [(2304, 2048), (3072, 2560), (4096, 2048), (1024, 4096), (4096, 1024), (1152, 2048), (512, 4096), (4096, 256), (640, 2560), (384, 4100)])
def test_kron_fp32_update(psgd, poisoned, M, N):
    rng = np.random.default_rng(M + 11 * N)
    Ql, Qr, dX, dG = _tri(rng, M, scale=1.5), _tri(rng, N), _randn(rng, M, N), _randn(rng, M, N, scale=2.0)
    _twice(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), poisoned)


@pytest.mark.parametrize("M,N", [(1024, 1024), (1024, 2304), (2048, 2048), (2100, 1500), (4096, 4096)])
def test_kron_bf16_operands(psgd, poisoned, M, N):
    rng = np.random.default_rng(M + 13 * N)
    Ql, Qr = _tri(rng, M, scale=1.5), _tri(rng, N)
    dX, dG = _randn(rng, M, N).bfloat16(), _randn(rng, M, N, scale=2.0).bfloat16()
    _twice(lambda: psgd.precond_grad_kron(Ql, Qr, dG), poisoned)
    _twice(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), poisoned)


def test_kron_sparse_formats(psgd, poisoned):
    M, N = 1024, 2048
    rng = np.random.default_rng(5)
    dX, dG = _randn(rng, M, N), _randn(rng, M, N, scale=2.0)
    Ql, Qr = _tri(rng, M), _tri(rng, N)
    ql = torch.cat([torch.ones(1, M), torch.zeros(1, M)]).cuda() + 0.01 * _randn(rng, 2, M)
    ql[1, M - 1] = 0.0
    qr = (1.0 + 0.1 * _randn(rng, 1, N)).abs()
    for L, R in ((Ql, qr), (ql, Qr), (ql, qr)):
        _twice(lambda: psgd.update_precond_kron(L, R, dX, dG, 0.01), poisoned, expect_ws=False)
        _twice(lambda: psgd.precond_grad_kron(L, R, dG), poisoned, expect_ws=False)


def test_kron_batched_small_layers(psgd, poisoned):
    rng = np.random.default_rng(9)
    shapes = [(26, 6), (151, 16), (401, 120), (121, 84), (85, 10)]                # BASELINE config 3 (LeNet5)
    Qls, Qrs = [_tri(rng, m, 0.05) for m, _ in shapes], [_tri(rng, n, 0.05) for _, n in shapes]
    dXs, dGs = [_randn(rng, m, n) for m, n in shapes], [_randn(rng, m, n, scale=2.0) for m, n in shapes]

    def upd():
        return [t for pair in psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01) for t in pair]
    _twice(upd, poisoned)
    _twice(lambda: psgd.precond_grad_kron_batched(Qls, Qrs, dGs), poisoned)


@pytest.mark.parametrize("N,r", [(1_000_000, 10), (3_000_017, 20), (500_000, 3), (400_000, 40), (300_000, 64)])
def test_uvd(psgd, poisoned, N, r):
    g = torch.Generator(device="cuda").manual_seed(N + r)
    scale = (1.0 / (N * r)) ** 0.5
    U0, V0 = torch.randn(N, r, device="cuda", generator=g) * scale, torch.randn(N, r, device="cuda", generator=g) * scale
    d0 = torch.ones(N, 1, device="cuda")
    v, grad = torch.randn(N, 1, device="cuda", generator=g), torch.randn(N, 1, device="cuda", generator=g)
    h = v * torch.exp(torch.empty(N, 1, device="cuda").uniform_(-2.0, 2.0, generator=g))
    _twice(lambda: psgd.precond_grad_UVd_math(U0, V0, d0, grad), poisoned)
    for update_u in (True, False):
        for balance in (False, True):

            def upd():
                U, V, d = U0.clone(), V0.clone(), d0.clone()
                psgd.update_precond_UVd_math_(U, V, d, v, h, 0.01, psgd._tiny, balance=balance, update_U=update_u)
                return U, V, d
            _twice(upd, poisoned, expect_ws=False)

            def fused():
                U, V, d = U0.clone(), V0.clone(), d0.clone()
                out = psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, grad, 0.01, psgd._tiny, balance=balance,
                                                                   update_U=update_u)
                return U, V, d, out
            _twice(fused, poisoned, expect_ws=False)


@pytest.mark.parametrize("N,r", [(1_000_000, 10), (400_000, 40)])
def test_sparse_lu(psgd, poisoned, N, r):
    from splu_cases import make_splu_problem
    q = {k: torch.from_numpy(v).cuda() for k, v in make_splu_problem(N, r, seed=N + r).items()}
    L12, l3, U12, u3, dx, dg = q["L12"], q["l3"], q["U12"], q["u3"], q["dx"], q["dg"]
    _twice(lambda: psgd.precond_grad_splu(L12, l3, U12, u3, [dg]), poisoned, expect_ws=False)
    _twice(lambda: psgd.update_precond_splu(L12, l3, U12, u3, [dx], [dg], 0.01), poisoned, expect_ws=False)


@pytest.fixture(scope="module")
def pg():
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29546")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("N,r", [(1_000_000, 10), (300_000, 40)])
def test_sharded_stages_on_a_one_rank_group(psgd, poisoned, pg, N, r):
    """the row-sharded driver (stage entry points + exchanges on the library's own RCCL communicator): its stage workspaces and gather
    buffers start poisoned too"""
    from psgd_tf_amd import sharded
    g = torch.Generator(device="cuda").manual_seed(N + r)
    scale = (1.0 / (N * r)) ** 0.5
    U0, V0 = torch.randn(N, r, device="cuda", generator=g) * scale, torch.randn(N, r, device="cuda", generator=g) * scale
    d0 = torch.ones(N, 1, device="cuda")
    v, grad = torch.randn(N, 1, device="cuda", generator=g), torch.randn(N, 1, device="cuda", generator=g)
    h = v * 1.5
    _twice(lambda: sharded.precond_grad_UVd_math(U0, V0, d0, grad), poisoned, expect_ws=False)
    for update_u in (True, False):
        def fused():
            U, V, d = U0.clone(), V0.clone(), d0.clone()
            out = sharded.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, grad, 0.01, psgd._tiny, balance=True, update_U=update_u)
            return U, V, d, out
        _twice(fused, poisoned, expect_ws=False)
