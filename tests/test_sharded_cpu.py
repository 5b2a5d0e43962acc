"""World-size-2 test of the row-sharded UVd driver (psgd_tf_amd/sharded.py) under gloo on CPU.

The product's stage backend is the HIP C ABI; here a NumPy stage backend (tests/cpu_stages.py)
is injected so that what is under test is the multi-rank choreography: the sharding, which
reduced buffer is all-reduced between which sweeps (SUM on fp64 sums, MAX on the fp32 max
buffer), and that both ranks take the same branches.  The concatenated shards must equal the
unsharded reference-order oracle."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import psgd_oracle as orc
from tests.uvd_cases import make_uvd_problem, rel_err

N, R, WORLD = 1003, 6, 2
TINY = float(np.finfo(np.float32).tiny)


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from psgd_tf_amd import sharded
        from tests.cpu_stages import NumpyStages
        calls = {"all_gather_into_tensor": 0, "all_reduce": 0, "broadcast": 0}
        for name in calls:                                  # count the collectives each sharded call issues
            def wrap(fn, name=name):
                def counted(*a, **k):
                    calls[name] += 1
                    return fn(*a, **k)
                return counted
            setattr(dist, name, wrap(getattr(dist, name)))
        snap = lambda: dict(calls)
        p = make_uvd_problem(N, R, seed=21, uv_gain=2.0, d_spread=0.3)
        p["U"] *= 5.0
        lo, hi = sharded.shard_rows(N, rank, world)
        t = {k: torch.from_numpy(p[k][lo:hi].astype(np.float64).copy()) for k in p}
        be = NumpyStages(R)
        # one update per branch combination, then an apply; branch bits: explicit, then broadcast from rank 0
        sharded.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY, balance=True,
                                         update_U=True, backend=be)
        sharded.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY, balance=False,
                                         update_U=False, backend=be)
        c_upd = snap()
        assert c_upd == {"all_gather_into_tensor": 5, "all_reduce": 0, "broadcast": 0}, c_upd   # 3 (balance) + 2
        gen = torch.Generator().manual_seed(1234 + rank)          # different seeds: rank 0's state must win
        sharded.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY, generator=gen,
                                         backend=be)
        assert calls["broadcast"] == 1                            # the one-time generator synchronisation ...
        gathers = calls["all_gather_into_tensor"]
        out = sharded.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"], backend=be)
        assert calls["all_gather_into_tensor"] - gathers == 2 and calls["all_reduce"] == 0
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), U=t["U"].numpy(), V=t["V"].numpy(), d=t["d"].numpy(),
                 out=out.numpy(), lo=lo, hi=hi)
        # fused update -> apply (one V pass less): both branches, continuing from that state
        outs_f = []
        for upd in (True, False):
            before = snap()
            outs_f.append(sharded.update_precond_UVd_math_and_precond_grad(
                t["U"], t["V"], t["d"], t["v"], t["h"], t["g"], 0.01, TINY, balance=False, update_U=upd, backend=be))
            assert calls["all_gather_into_tensor"] - before["all_gather_into_tensor"] == 2      # Gram, [4r sums | max]
            assert calls["all_reduce"] == 0 and calls["broadcast"] == before["broadcast"]
        # ... and later drawn calls with the same generator object exchange nothing for the branches
        before = snap()
        sharded.update_precond_UVd_math_(t["U"].clone(), t["V"].clone(), t["d"].clone(), t["v"], t["h"], 0.01, TINY,
                                         generator=gen, backend=be)
        assert calls["broadcast"] == before["broadcast"]
        np.savez(os.path.join(outdir, "fused%d.npz" % rank), U=t["U"].numpy(), V=t["V"].numpy(), d=t["d"].numpy(),
                 o0=outs_f[0].numpy(), o1=outs_f[1].numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
def test_sharded_update_and_apply_equal_unsharded_oracle():
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(WORLD, _free_port(), outdir), nprocs=WORLD, join=True)
        parts = [np.load(os.path.join(outdir, "rank%d.npz" % k)) for k in range(WORLD)]
        fused = [np.load(os.path.join(outdir, "fused%d.npz" % k)) for k in range(WORLD)]
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == N
    got = {k: np.concatenate([q[k] for q in parts], 0) for k in ("U", "V", "d", "out")}

    p = make_uvd_problem(N, R, seed=21, uv_gain=2.0, d_spread=0.3)
    p["U"] *= 5.0
    q = {k: p[k].astype(np.float64) for k in p}
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY, balance=True, update_U=True)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY, balance=False, update_U=False)
    # third call: branches drawn by rank 0 from Generator(1234): reproduce the draw order (:562 then :588)
    gen = torch.Generator().manual_seed(1234)
    bal = bool(torch.rand((), generator=gen).item() < 0.01)
    upd = bool(torch.rand((), generator=gen).item() < 0.5)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY, balance=bal, update_U=upd)
    ref_out = orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])
    # the fp32 MAX buffers round max|.| to fp32 (as the product does), hence 1e-7 rather than 1e-13
    for k in ("U", "V", "d"):
        assert rel_err(got[k], q[k]) < 1e-7, k
    assert rel_err(got["out"], ref_out) < 1e-7
    # the fused calls continue from that state: update (U branch) + apply, then update (V branch) + apply
    gotf = {k: np.concatenate([f[k] for f in fused], 0) for k in ("U", "V", "d", "o0", "o1")}
    refs = []
    for upd in (True, False):
        orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY, balance=False, update_U=upd)
        refs.append(orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"]))
    assert rel_err(gotf["o0"], refs[0]) < 1e-7 and rel_err(gotf["o1"], refs[1]) < 1e-7
    for k in ("U", "V", "d"):
        assert rel_err(gotf[k], q[k]) < 1e-7, k


def test_shard_rows_cover_and_align():
    from psgd_tf_amd import sharded
    for n, w in [(100_000_000, 8), (1003, 2), (7, 4), (64, 8), (1, 2)]:
        edges = [sharded.shard_rows(n, k, w) for k in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        for (lo, hi), (lo2, _) in zip(edges, edges[1:]):
            assert hi == lo2 and lo <= hi
        assert all(lo % 64 == 0 for lo, hi in edges if hi > lo)


# ----------------------------------------------------------------------------- sparse LU, tail rows sharded
SPLU_N, SPLU_R = 61, 5


def _splu_local(p, lo, hi):
    """This rank's tensors: the r x r corner replicated, tail rows [lo, hi) (tail-relative)."""
    r = SPLU_R
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float64)))
    loc = {"L12": f(np.concatenate([p["L12"][:r], p["L12"][r + lo:r + hi]], 0)),
           "U12": f(np.concatenate([p["U12"][:, :r], p["U12"][:, r + lo:r + hi]], 1)),
           "l3": f(p["l3"][lo:hi]), "u3": f(p["u3"][lo:hi])}
    for k in ("dx", "dg", "g"):
        loc[k] = f(np.concatenate([p[k][:r], p[k][r + lo:r + hi]], 0))
    return loc


def _splu_worker(rank, world, port, outdir, cut):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from psgd_tf_amd import sharded
        from tests.cpu_stages import NumpySpluStages
        from tests.splu_cases import make_splu_problem
        p = make_splu_problem(SPLU_N, SPLU_R, seed=33)
        n2 = SPLU_N - SPLU_R
        lo, hi = (0, cut) if rank == 0 else (cut, n2)
        t = _splu_local(p, lo, hi)
        be = NumpySpluStages(SPLU_R)
        out0 = sharded.precond_grad_splu(t["L12"], t["l3"], t["U12"], t["u3"], t["g"], backend=be)
        new = sharded.update_precond_splu(t["L12"], t["l3"], t["U12"], t["u3"], t["dx"], t["dg"], 0.1, TINY, backend=be)
        out1 = sharded.precond_grad_splu(*new, t["g"], backend=be)
        np.savez(os.path.join(outdir, "splu%d.npz" % rank), out0=out0.numpy(), out1=out1.numpy(),
                 L12=new[0].numpy(), l3=new[1].numpy(), U12=new[2].numpy(), u3=new[3].numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("cut", [29, 56])          # 29 + 27 rows; 56 + 0: one rank holds the corner only
def test_sharded_splu_equals_unsharded_oracle(cut):
    from tests.splu_cases import make_splu_problem
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_splu_worker, args=(WORLD, _free_port(), outdir, cut), nprocs=WORLD, join=True)
        parts = [np.load(os.path.join(outdir, "splu%d.npz" % k)) for k in range(WORLD)]
    r = SPLU_R
    p = {k: v.astype(np.float64) for k, v in make_splu_problem(SPLU_N, SPLU_R, seed=33).items()}
    ref0 = orc.precond_grad_splu(p["L12"], p["l3"], p["U12"], p["u3"], [p["g"]])[0]
    new = orc.update_precond_splu(p["L12"], p["l3"], p["U12"], p["u3"], [p["dx"]], [p["dg"]], 0.1)
    ref1 = orc.precond_grad_splu(*new, [p["g"]])[0]
    asm_vec = lambda key: np.concatenate([parts[0][key][:r], parts[0][key][r:], parts[1][key][r:]], 0)
    for a, b in ((parts[0]["out0"][:r], parts[1]["out0"][:r]), (parts[0]["L12"][:r], parts[1]["L12"][:r]),
                 (parts[0]["U12"][:, :r], parts[1]["U12"][:, :r])):
        assert np.array_equal(a, b)                     # replicated corner results are identical on both ranks
    # the fp32 MAX buffer rounds the maxima to fp32 (as the product does), hence 1e-7 rather than 1e-13
    assert rel_err(asm_vec("out0"), ref0) < 1e-12
    L12 = np.concatenate([parts[0]["L12"], parts[1]["L12"][r:]], 0)
    U12 = np.concatenate([parts[0]["U12"], parts[1]["U12"][:, r:]], 1)
    l3, u3 = np.concatenate([parts[0]["l3"], parts[1]["l3"]], 0), np.concatenate([parts[0]["u3"], parts[1]["u3"]], 0)
    for got, want in ((L12, new[0]), (l3, new[1]), (U12, new[2]), (u3, new[3])):
        assert rel_err(got, want) < 1e-7
    assert rel_err(asm_vec("out1"), ref1) < 1e-7


# ----------------------------------------------------------------------------- class UVd, row-sharded (UVd(..., group=pg))
CLS_SHAPES = [(3, 5), (7,), (4, 4), (6, 1), (1,)]          # rank 0 owns the first two tensors (22 rows), rank 1 the rest (23)
CLS_SPLIT, CLS_R, CLS_STEPS = 2, 3, 5


def _cls_setup():
    """global parameters, initial state and the probe vectors of every step (seeded; the same in every process)"""
    g = torch.Generator().manual_seed(77)
    params = [torch.randn(s, generator=g) * 0.7 for s in CLS_SHAPES]
    n = sum(p.numel() for p in params)
    sc = (1.0 / (n * CLS_R)) ** 0.5
    U, V = torch.randn(n, CLS_R, generator=g) * sc * 6, torch.randn(n, CLS_R, generator=g) * sc * 6
    d = torch.exp(torch.randn(n, 1, generator=g) * 0.2)
    probes = [[torch.randn(s, generator=g) for s in CLS_SHAPES] for _ in range(CLS_STEPS)]
    return params, U, V, d, probes


def _cls_loss(ps, salt):
    """a rank's loss on ITS parameters (the global loss is the sum over ranks: block-diagonal Hessian)"""
    flat = torch.cat([p.reshape(-1) for p in ps])
    w = torch.cos(torch.arange(flat.numel(), dtype=flat.dtype) * 0.37 + salt)
    return 0.5 * torch.sum((1.0 + w * w) * flat * flat) + 0.25 * torch.sum(flat ** 4) + torch.sum(w * flat) ** 2


def _cls_schedule(opt, it):
    """hyper-parameters changed between steps (psgd.py:673-680 are mutable): clip on / off, exact / finite-difference Hv,
    and a step that leaves the preconditioner alone"""
    opt.grad_clip_max_norm.assign([0.05, float("inf"), 0.05, 0.05, float("inf")][it])
    opt.exact_hessian_vector_product.assign([True, True, False, True, False][it])
    opt.preconditioner_update_probability.assign([1.0, 1.0, 1.0, 0.0, 1.0][it])


def _cls_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import preconditioned_stochastic_gradient_descent as psgd
        from psgd_tf_amd import preconditioned_stochastic_gradient_descent as prod
        from tests.cpu_stages import NumpyStages
        params, U, V, d, probes = _cls_setup()
        mine = slice(0, CLS_SPLIT) if rank == 0 else slice(CLS_SPLIT, None)
        own = [p.clone().requires_grad_(True) for p in params[mine]]
        lo = sum(p.numel() for p in params[:mine.start or 0])
        hi = lo + sum(p.numel() for p in own)
        gen = torch.Generator().manual_seed(4321 + 17 * rank)            # different seeds: rank 0's coins must win
        opt = psgd.UVd(own, rank_of_modification=CLS_R, lr_params=0.004, lr_preconditioner=0.1, generator=gen,
                       group=dist.group.WORLD, stage_backend=NumpyStages(CLS_R, dtype=np.float32))
        assert opt._num_params_global == sum(p.numel() for p in params) and opt._U.shape == (hi - lo, CLS_R)
        assert abs(float(opt._U.std()) / (1.0 / (opt._num_params_global * CLS_R)) ** 0.5 - 1.0) < 0.4   # :687 on the GLOBAL N
        opt._U.copy_(U[lo:hi]); opt._V.copy_(V[lo:hi]); opt._d.copy_(d[lo:hi])
        calls = {"all_gather_into_tensor": 0, "all_reduce": 0, "broadcast": 0}
        for name in calls:
            def wrap(fn, name=name):
                def counted(*a, **k):
                    calls[name] += 1
                    return fn(*a, **k)
                return counted
            setattr(dist, name, wrap(getattr(dist, name)))
        per_step = []
        for it in range(CLS_STEPS):
            queue = {id(p): q for p, q in zip(own, probes[it][mine])}
            prod._randn_like = lambda p: queue[id(p)].clone()
            _cls_schedule(opt, it)
            before = dict(calls)
            opt.step(lambda: _cls_loss(own, float(rank)))
            per_step.append({k: calls[k] - before[k] for k in calls})
        # collectives per step: 2 exchanges (+1 scalar all-reduce when clipping), nothing else -- no broadcast after set-up
        clip = [True, False, True, True, False]
        for it, c in enumerate(per_step):
            assert c == {"all_gather_into_tensor": 2, "all_reduce": int(clip[it]), "broadcast": 0}, (it, c)
            assert sum(c.values()) <= 3
        np.savez(os.path.join(outdir, "cls%d.npz" % rank), U=opt._U.numpy(), V=opt._V.numpy(), d=opt._d.numpy(),
                 p=torch.cat([p.detach().reshape(-1) for p in own]).numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_uvd_class_equals_unsharded_oracle_and_counts_collectives():
    """UVd(..., group=pg) on two ranks (psgd.py:692-764 row-sharded): parameters and state after five steps -- clip on and off,
    exact and finite-difference Hv, one step without a preconditioner update -- against oracle.uvd_step on the global vector;
    at most three collectives per step."""
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_cls_worker, args=(WORLD, _free_port(), outdir), nprocs=WORLD, join=True)
        sh = [np.load(os.path.join(outdir, "cls%d.npz" % k)) for k in range(WORLD)]
    got = {k: np.concatenate([s[k] for s in sh], 0) for k in ("U", "V", "d", "p")}
    params, U, V, d, probes = _cls_setup()
    U, V, d = (x.numpy().astype(np.float64) for x in (U, V, d))
    groups = [slice(0, CLS_SPLIT), slice(CLS_SPLIT, None)]
    gen = torch.Generator().manual_seed(4321)                            # rank 0's generator
    tiny, delta = float(np.finfo(np.float32).tiny), float(np.finfo(np.float32).eps) ** 0.5

    class Hyp:                                                           # the schedule, replayed on plain attributes
        def __init__(self): self.value = None
        def assign(self, v): self.value = v
    hy = type("H", (), {})()
    hy.grad_clip_max_norm, hy.exact_hessian_vector_product, hy.preconditioner_update_probability = Hyp(), Hyp(), Hyp()

    def grads_of(ps):                                                    # per rank, exactly as the workers compute them (fp32 autograd)
        out = []
        for k, sl in enumerate(groups):
            own = [p.clone().requires_grad_(True) for p in ps[sl]]
            out.append((own, torch.autograd.grad(_cls_loss(own, float(k)), own, create_graph=True)))
        return out
    for it in range(CLS_STEPS):
        _cls_schedule(hy, it)
        exact = hy.exact_hessian_vector_product.value
        update_Q = bool(torch.rand((), generator=gen).item() < hy.preconditioner_update_probability.value)   # :703
        bal = upd = False
        vs = Hvs = None
        if update_Q:
            gs = grads_of(params)
            if exact:
                vs = probes[it]
                Hvs = [h for (own, g), sl in zip(gs, groups) for h in torch.autograd.grad(g, own, vs[sl])]
                stepped_from = params
            else:
                vs = [q * np.float32(delta) for q in probes[it]]
                pert = [p + v for p, v in zip(params, vs)]
                pg = [x.detach() for own, g in grads_of(pert) for x in g]
                Hvs = [a - b.detach() for a, b in zip(pg, [x for own, g in gs for x in g])]
                stepped_from = pert
            bal = bool(torch.rand((), generator=gen).item() < 0.01)      # :562
            upd = bool(torch.rand((), generator=gen).item() < 0.5)       # :588
        else:
            gs = grads_of(params)
            stepped_from = params
        grads = [x.detach() for own, g in gs for x in g]
        f64 = lambda ts: [t.detach().numpy().astype(np.float64) for t in ts] if ts is not None else None
        new = orc.uvd_step(f64(stepped_from), f64(grads), f64(Hvs), f64(vs), U, V, d, 0.004, 0.1,
                           hy.grad_clip_max_norm.value, tiny, balance=bal, update_U=upd, update_Q=update_Q, exact=exact,
                           delta_param_scale=delta)
        params = [torch.from_numpy(x.astype(np.float32)) for x in new]
    want_p = np.concatenate([p.reshape(-1).numpy() for p in params])
    assert rel_err(got["p"], want_p) < 2e-5
    for k, ref in (("U", U), ("V", V), ("d", d)):
        assert rel_err(got[k], ref) < 2e-5, k
