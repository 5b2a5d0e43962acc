"""GPU: the UVd state's allocator (psgd_tf_amd/placement.py).  Placement changes addresses only -- every result must be the
same BITS as with plain allocations -- and the probe must come back with a working arena whatever it finds."""
import numpy as np
import pytest
import torch

from tests.uvd_cases import TINY32, make_uvd_problem

gpu = pytest.mark.gpu


def _plain(p, dev):
    return {k: torch.from_numpy(v).to(dev) for k, v in p.items()}


def _into(arena, p):
    for k, name in (("U", "U"), ("V", "V"), ("d", "d"), ("g", "g"), ("v", "v"), ("h", "h")):
        getattr(arena, name).copy_(torch.from_numpy(p[k]))


@gpu
@pytest.mark.parametrize("N,r", [(300_007, 20), (64_000, 10), (1_000_003, 32)])
def test_packed_arena_gives_the_same_bits(hip_lib, N, r):
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import placement
    dev = torch.device("cuda:0")
    p = make_uvd_problem(N, r, seed=3, uv_gain=2.0, d_spread=0.2)
    a = _plain(p, dev)
    arena = placement.UVdArena.packed(N, r, dev)
    assert all(getattr(arena, k).data_ptr() % 256 == 0 for k in ("U", "V", "d", "out", "ws", "g", "v", "h"))
    _into(arena, p)
    arena.install_workspace()
    for upd in (True, False):
        want = psgd.update_precond_UVd_math_and_precond_grad(a["U"], a["V"], a["d"], a["v"], a["h"], a["g"], 0.01, TINY32,
                                                             balance=False, update_U=upd)
        got = psgd.update_precond_UVd_math_and_precond_grad(arena.U, arena.V, arena.d, arena.v, arena.h, arena.g, 0.01, TINY32,
                                                            balance=False, update_U=upd, out=arena.out)
        assert got.data_ptr() == arena.out.data_ptr()
        assert torch.equal(got, want)
        for k in ("U", "V", "d"):
            assert torch.equal(getattr(arena, k), a[k]), k
    # the two reference-named calls on the placed state, and the workspace the arena installed is the one they use
    psgd.update_precond_UVd_math_(arena.U, arena.V, arena.d, arena.v, arena.h, 0.01, TINY32, balance=True, update_U=True)
    psgd.update_precond_UVd_math_(a["U"], a["V"], a["d"], a["v"], a["h"], 0.01, TINY32, balance=True, update_U=True)
    ref = psgd.precond_grad_UVd_math(a["U"], a["V"], a["d"], a["g"])
    assert torch.equal(psgd.precond_grad_UVd_math(arena.U, arena.V, arena.d, arena.g), ref)
    placed = psgd.precond_grad_UVd_math(arena.U, arena.V, arena.d, arena.g, out=arena.out)       # the result written into the arena
    assert placed.data_ptr() == arena.out.data_ptr() and torch.equal(placed, ref)
    assert psgd.uvd_workspace(dev, N, r).data_ptr() == arena.ws.data_ptr()
    with pytest.raises(ValueError):
        psgd.update_precond_UVd_math_and_precond_grad(arena.U, arena.V, arena.d, arena.v, arena.h, arena.g, 0.01, TINY32,
                                                      balance=False, update_U=True, out=arena.out.view(-1)[:-1])


@gpu
def test_probe_returns_a_working_arena(hip_lib):
    """The probe on a 2M-row problem with 64-MiB thin buffers and three tries: whatever it finds (at this size the written streams
    are cache-resident, so usually "no second region" and the packed arena again), the arena it returns computes the same bits
    as plain allocations, and the log holds the packed timing and the region search."""
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import placement
    dev = torch.device("cuda:0")
    N, r = 2_000_000, 20
    log = []
    arena = placement.UVdArena.probe(N, r, dev, max_tries=3, chunk_bytes=64 << 20, log=log)
    assert arena.info["layout"] == "packed" or arena.info["layout"].startswith("two regions")
    assert log[0]["layout"] == "packed" and log[0]["step_U_ms"] > 0
    assert any("thin_buffer_ms" in rec and rec["same_buffer_ms"] > 0 and 1 <= len(rec["thin_buffer_ms"]) <= 3 for rec in log)
    assert arena.bytes_held >= sum(placement.UVdArena.region_bytes(N, r).values())
    p = make_uvd_problem(N, r, seed=5)
    a = _plain(p, dev)
    _into(arena, p)
    arena.install_workspace()
    want = psgd.update_precond_UVd_math_and_precond_grad(a["U"], a["V"], a["d"], a["v"], a["h"], a["g"], 0.01, TINY32,
                                                         balance=False, update_U=True)
    got = psgd.update_precond_UVd_math_and_precond_grad(arena.U, arena.V, arena.d, arena.v, arena.h, arena.g, 0.01, TINY32,
                                                        balance=False, update_U=True, out=arena.out)
    assert torch.equal(got, want) and torch.equal(arena.U, a["U"]) and torch.equal(arena.d, a["d"])


@gpu
def test_two_buffer_arena_gives_the_same_bits(hip_lib):
    """the layout the probe builds when it finds a second region: U, V in one buffer, the thin streams in another"""
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import placement
    dev = torch.device("cuda:0")
    N, r = 500_003, 20
    fac = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
    thin = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    arena = placement.UVdArena.two_buffers(N, r, dev, fac, thin)
    assert arena.U.data_ptr() == fac.data_ptr() and arena.d.data_ptr() == thin.data_ptr() and arena.bytes_held == (160 << 20)
    p = make_uvd_problem(N, r, seed=8, uv_gain=2.0)
    a = _plain(p, dev)
    _into(arena, p)
    arena.install_workspace()
    for upd in (True, False):
        want = psgd.update_precond_UVd_math_and_precond_grad(a["U"], a["V"], a["d"], a["v"], a["h"], a["g"], 0.01, TINY32,
                                                             balance=False, update_U=upd)
        got = psgd.update_precond_UVd_math_and_precond_grad(arena.U, arena.V, arena.d, arena.v, arena.h, arena.g, 0.01, TINY32,
                                                            balance=False, update_U=upd, out=arena.out)
        assert torch.equal(got, want) and torch.equal(arena.V, a["V"])
    with pytest.raises(ValueError):
        placement.UVdArena.two_buffers(N, r, dev, fac[:1 << 20], thin)


def test_arena_layouts():
    """host logic of the layouts (the region sizes come from the library): regions one after the other, 256-byte aligned, no
    overlap; the factor buffer of the probe is the power of two that holds U and V"""
    from psgd_tf_amd import placement
    N, r = 100_000_000, 20
    sz = placement.UVdArena.region_bytes(N, r)
    off, end = placement.UVdArena.sequential(N, r, placement.UVdArena.NAMES)
    spans = sorted((off[k], off[k] + sz[k]) for k in off)
    assert spans[0][0] == 0 and all(a[1] <= b[0] < a[1] + 256 for a, b in zip(spans, spans[1:])) and spans[-1][1] <= end
    assert all(o % 256 == 0 for o in off.values()) and set(off) == set(sz)
    off2, end2 = placement.UVdArena.sequential(N, r, placement.UVdArena.THIN, start=1000)
    assert min(off2.values()) == 1024 and end2 - 1024 < (4 << 30)              # the thin streams of the headline fit a 4-GiB buffer
    assert 2 * sz["U"] <= (16 << 30)                                           # and its factors a 16-GiB one


@gpu
def test_uvd_class_with_a_placed_state(hip_lib):
    """UVd(..., placement='packed'): same parameters and state after three steps as the plain optimizer, bit for bit."""
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import preconditioned_stochastic_gradient_descent as prod
    dev = torch.device("cuda:0")
    shapes = [(40, 30), (1000,), (17, 1)]
    g0 = torch.Generator().manual_seed(11)
    init = [(torch.randn(s, generator=g0) * 0.3).to(dev) for s in shapes]
    probes = [[torch.randn(s, generator=g0).to(dev) for s in shapes] for _ in range(3)]
    n = sum(p.numel() for p in init)
    U0, V0 = (torch.randn(n, 10, generator=g0) * 0.02).to(dev), (torch.randn(n, 10, generator=g0) * 0.02).to(dev)

    def loss(ps):
        flat = torch.cat([p.reshape(-1) for p in ps])
        return 0.5 * torch.sum(flat * flat) + 0.25 * torch.sum(flat ** 4) + 0.01 * torch.sum(flat) ** 2
    keep = prod._randn_like
    res = []
    try:
        for placement_mode in (None, "packed"):
            ps = [p.clone().requires_grad_(True) for p in init]
            opt = psgd.UVd(ps, rank_of_modification=10, lr_params=0.01, lr_preconditioner=0.05, grad_clip_max_norm=0.5,
                           generator=torch.Generator().manual_seed(3), placement=placement_mode)
            assert (opt._arena is not None) == (placement_mode is not None)
            opt._U.copy_(U0); opt._V.copy_(V0)
            for it in range(3):
                queue = {id(p): q for p, q in zip(ps, probes[it])}
                prod._randn_like = lambda p: queue[id(p)].clone()
                opt.exact_hessian_vector_product.assign(it != 1)
                opt.step(lambda: loss(ps))
            res.append([p.detach().clone() for p in ps] + [opt._U.clone(), opt._V.clone(), opt._d.clone()])
    finally:
        prod._randn_like = keep
    for a, b in zip(*res):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        psgd.UVd([init[0].clone().requires_grad_(True)], placement="best")


@gpu
def test_functional_example_converges(hip_lib):
    """examples/uvd_functional_step.py: the functional API on a placed state; the preconditioner learns the inverse curvatures of a
    quadratic whose curvatures span a factor of 100 and the loss falls by more than 1e3 in 200 steps"""
    from examples.uvd_functional_step import run
    losses, arena = run(N=300_000, r=10, steps=200, mode="packed")
    assert all(np.isfinite(losses)) and losses[-1] < 1e-3 * losses[0], (losses[0], losses[-1])
    assert arena.info["layout"] == "packed"
