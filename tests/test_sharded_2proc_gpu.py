"""GPU, two PROCESSES: the sharded drivers with the product's HIP stage kernels on each rank's row shard and REAL
collectives between the processes.  Two transports:
  gloo-one-device   both ranks on cuda:0, gloo over device tensors (RCCL refuses two ranks on one GPU): what a 1-GPU box can run
  rccl-two-devices  rank k on cuda:k, backend "nccl" (= RCCL over xGMI): runs wherever torch.cuda.device_count() >= 2, skipped
                    otherwise -- the first multi-GPU box exercises RCCL here, not only through bench.py
Complements tests/test_sharded_gpu.py (collectives emulated inside one process) and tests/test_sharded_cpu.py (real
collectives, NumPy stage kernels): here both are real.  Rank 0 compares the concatenated shards with the unsharded HIP call
and the oracle, and the REDUCED workspace regions of the two ranks bit for bit (the property the all-gather + rank-order fold
exists for: every rank redoes the r x r algebra on identical numbers)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

from oracle import psgd_oracle as orc
from tests.uvd_cases import TINY32, make_uvd_problem, rel_err
from tests.splu_cases import make_splu_problem

pytestmark = pytest.mark.gpu
WORLD = 2


def _worker(rank, port, outdir, N, r, two_devices):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dev = torch.device("cuda", rank if two_devices else 0)
    torch.cuda.set_device(dev)
    if two_devices:
        dist.init_process_group("nccl", rank=rank, world_size=WORLD, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from psgd_tf_amd import _lib, sharded
    p = make_uvd_problem(N, r, seed=21, uv_gain=2.0, d_spread=0.3)
    lo, hi = sharded.shard_rows(N, rank, WORLD)
    t = {k: torch.from_numpy(np.ascontiguousarray(v[lo:hi])).to(dev) for k, v in p.items()}
    red = {}                                      # reduced workspace regions, snapshot right after the call that made them
    be = sharded.hip_backend_for(t["U"]) if r <= _lib.UVD_MAX_RANK else None      # (r > 32: the wide path all-reduces tensors)
    for upd in (True, False):
        sharded.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY32, balance=False, update_U=upd)
        if be is not None:
            red["upd%d_gram" % upd] = be.sums(11).cpu().numpy().copy()
            red["upd%d_max" % upd] = be.maxbuf(12).cpu().numpy().copy()
    out = sharded.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
    if be is not None:
        red["apply_s2"] = be.sums(2).cpu().numpy().copy()
    ex0 = sharded.EXCHANGES["count"]
    outf = sharded.update_precond_UVd_math_and_precond_grad(t["U"], t["V"], t["d"], t["v"], t["h"], t["g"], 0.01, TINY32,
                                                            balance=False, update_U=True)
    fused_exchanges = sharded.EXCHANGES["count"] - ex0
    if be is not None:
        red["fused_gram"] = be.sums(11).cpu().numpy().copy()
        red["fused_s13"] = be.sums(13).cpu().numpy().copy()
    # sparse LU: the r x r corner replicated, the tail rows split between the ranks (tests/test_sharded_cpu.py layout)
    n, rr = N // 4, (7 if r <= 32 else 40)                  # (rank 40: column chunks of L2 / U2' with an all-reduce per exchange)
    s = make_splu_problem(n, rr, seed=5)
    n2 = n - rr
    cut = (n2 // 2) // 64 * 64
    lo2, hi2 = (0, cut) if rank == 0 else (cut, n2)
    f = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    L12 = f(np.concatenate([s["L12"][:rr], s["L12"][rr + lo2:rr + hi2]], 0))
    U12 = f(np.concatenate([s["U12"][:, :rr], s["U12"][:, rr + lo2:rr + hi2]], 1))
    l3, u3 = f(s["l3"][lo2:hi2]), f(s["u3"][lo2:hi2])
    loc = {k: f(np.concatenate([s[k][:rr], s[k][rr + lo2:rr + hi2]], 0)) for k in ("dx", "dg", "g")}
    pre0 = sharded.precond_grad_splu(L12, l3, U12, u3, loc["g"])
    new = sharded.update_precond_splu(L12, l3, U12, u3, loc["dx"], loc["dg"], 0.1, TINY32)
    sb = sharded._splu_backend_for(L12) if rr <= _lib.UVD_MAX_RANK else None
    if sb is not None:
        red["splu_s3"] = sb.sums(3).cpu().numpy().copy()
        red["splu_max"] = sb.maxbuf().cpu().numpy().copy()
    pre1 = sharded.precond_grad_splu(*new, loc["g"])
    if sb is not None:
        red["splu_apply_s2"] = sb.sums(2).cpu().numpy().copy()
    np.savez(os.path.join(outdir, "r%d.npz" % rank), U=t["U"].cpu().numpy(), V=t["V"].cpu().numpy(), d=t["d"].cpu().numpy(),
             out=out.cpu().numpy(), outf=outf.cpu().numpy(), pre0=pre0.cpu().numpy(), pre1=pre1.cpu().numpy(),
             L12=new[0].cpu().numpy(), l3=new[1].cpu().numpy(), U12=new[2].cpu().numpy(), u3=new[3].cpu().numpy(),
             backend=np.array(dist.get_backend()), device=np.array(dev.index), fused_exchanges=np.array(fused_exchanges),
             **{"red_" + k: v for k, v in red.items()})
    dist.barrier()
    dist.destroy_process_group()


def _two_devices():
    return torch.cuda.device_count() >= 2          # (counting devices does not initialise the GPU runtime)


@pytest.mark.parametrize("transport", [
    "gloo-one-device",
    pytest.param("rccl-two-devices", marks=pytest.mark.skipif(not _two_devices(), reason="needs two GPUs (rank k on cuda:k over RCCL)"))])
@pytest.mark.parametrize("N,r", [(200003, 20), (60003, 40)])      # r = 40: the wide-rank path (column chunks), sharded
def test_two_processes_real_kernels_real_collectives(hip_lib, N, r, transport):
    import torch.multiprocessing as mp
    import preconditioned_stochastic_gradient_descent as psgd
    outdir = tempfile.mkdtemp()
    port = 29600 + os.getpid() % 300
    two = transport == "rccl-two-devices"
    mp.start_processes(_worker, args=(port, outdir, N, r, two), nprocs=WORLD, join=True, start_method="spawn")
    sh = [np.load(os.path.join(outdir, "r%d.npz" % k)) for k in range(WORLD)]
    assert str(sh[0]["backend"]) == ("nccl" if two else "gloo")
    assert [int(s["device"]) for s in sh] == ([0, 1] if two else [0, 0])
    assert [int(s["fused_exchanges"]) for s in sh] == [2, 2]      # the fused step: two exchanges at every rank (r = 40 too, round 6)
    # every reduced region is the same BITS on both ranks (fold of the same copies in the same order)
    reds = [k for k in sh[0].files if k.startswith("red_")]
    assert ("red_fused_s13" in reds) == (r <= 32) and ("red_splu_s3" in reds) == (r <= 32)
    for k in reds:
        assert sh[0][k].tobytes() == sh[1][k].tobytes(), k
    got = {k: np.concatenate([s[k] for s in sh], 0) for k in ("U", "V", "d", "out", "outf")}
    p = make_uvd_problem(N, r, seed=21, uv_gain=2.0, d_spread=0.3)
    a = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    q = {k: v.astype(np.float64) for k, v in p.items()}
    for upd in (True, False):
        psgd.update_precond_UVd_math_(a["U"], a["V"], a["d"], a["v"], a["h"], 0.01, TINY32, balance=False, update_U=upd)
        orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=upd)
    out = psgd.precond_grad_UVd_math(a["U"], a["V"], a["d"], a["g"])
    assert rel_err(got["out"], orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < 1e-5
    assert rel_err(got["out"], out.cpu().numpy()) < 1e-5
    outf = psgd.update_precond_UVd_math_and_precond_grad(a["U"], a["V"], a["d"], a["v"], a["h"], a["g"], 0.01, TINY32,
                                                         balance=False, update_U=True)
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=True)
    for k in ("U", "V", "d"):
        assert rel_err(got[k], q[k]) < 1e-5, k
        assert rel_err(got[k], a[k].cpu().numpy()) < 1e-5, k
    assert rel_err(got["outf"], orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < 1e-5
    assert rel_err(got["outf"], outf.cpu().numpy()) < 1e-5
    # sparse LU: apply, update, apply with the updated factors
    n, rr = N // 4, (7 if r <= 32 else 40)
    s = make_splu_problem(n, rr, seed=5)
    s64 = {k: v.astype(np.float64) for k, v in s.items()}
    cat = lambda k, ax=0: np.concatenate([sh[0][k], sh[1][k][rr:] if ax == 0 else sh[1][k][:, rr:]], ax)
    want0 = orc.precond_grad_splu(s64["L12"], s64["l3"], s64["U12"], s64["u3"], [s64["g"]])[0]
    assert rel_err(cat("pre0"), want0) < 1e-5
    new = orc.update_precond_splu(s64["L12"], s64["l3"], s64["U12"], s64["u3"], [s64["dx"]], [s64["dg"]], 0.1)
    assert rel_err(cat("L12"), new[0]) < 1e-5 and rel_err(cat("U12", 1), new[2]) < 1e-5
    assert rel_err(np.concatenate([sh[0]["l3"], sh[1]["l3"]], 0), new[1]) < 1e-5
    assert rel_err(np.concatenate([sh[0]["u3"], sh[1]["u3"]], 0), new[3]) < 1e-5
    assert rel_err(cat("pre1"), orc.precond_grad_splu(*new, [s64["g"]])[0]) < 1e-5
    assert np.array_equal(sh[0]["L12"][:rr], sh[1]["L12"][:rr]) and np.array_equal(sh[0]["U12"][:, :rr], sh[1]["U12"][:, :rr])


# ----------------------------------------------------------------------------- class UVd, row-sharded: UVd(..., group=pg)
CLS_SHAPES = [(300, 40), (5000,), (64, 64), (1000, 1), (1,)]     # rank 0 owns the first two tensors (17000 rows), rank 1 the rest (5097)
CLS_SPLIT, CLS_R, CLS_STEPS = 2, 10, 5
CLS_CLIP = [0.05, float("inf"), 0.05, 0.05, float("inf")]        # psgd.py:675-678 (mutable between steps)
CLS_EXACT = [True, True, True, False, False]                     # :680; the finite-difference steps come last (see the test)
CLS_PROB = [1.0, 1.0, 0.0, 1.0, 1.0]                             # :679; step 2 leaves the preconditioner alone (:737-744)


def _cls_setup(dev):
    g = torch.Generator().manual_seed(91)
    params = [(torch.randn(s, generator=g) * 0.3).to(dev) for s in CLS_SHAPES]
    n = sum(p.numel() for p in params)
    sc = (1.0 / (n * CLS_R)) ** 0.5
    U, V = (torch.randn(n, CLS_R, generator=g) * sc * 6).to(dev), (torch.randn(n, CLS_R, generator=g) * sc * 6).to(dev)
    d = torch.exp(torch.randn(n, 1, generator=g) * 0.2).to(dev)
    probes = [[torch.randn(s, generator=g).to(dev) for s in CLS_SHAPES] for _ in range(CLS_STEPS)]
    return params, U, V, d, probes


def _cls_loss(ps, salt):
    """one rank's loss on ITS parameters; the global loss is the sum over the ranks (block-diagonal Hessian)"""
    flat = torch.cat([p.reshape(-1) for p in ps])
    w = torch.cos(torch.arange(flat.numel(), dtype=flat.dtype, device=flat.device) * 0.37 + salt)
    return 0.5 * torch.sum((1.0 + w * w) * flat * flat) + 0.25 * torch.sum(flat ** 4) + 1e-3 * torch.sum(w * flat) ** 2


def _cls_run(opt, prod, own_groups, probes_of, loss_fn, snap):
    for it in range(CLS_STEPS):
        queue = {id(p): q for ps, qs in zip(own_groups, probes_of(it)) for p, q in zip(ps, qs)}
        prod._randn_like = lambda p: queue[id(p)].clone()
        opt.grad_clip_max_norm.assign(CLS_CLIP[it])
        opt.exact_hessian_vector_product.assign(CLS_EXACT[it])
        opt.preconditioner_update_probability.assign(CLS_PROB[it])
        opt.step(loss_fn)
        snap(it)


def _cls_worker(rank, port, outdir, two_devices):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dev = torch.device("cuda", rank if two_devices else 0)
    torch.cuda.set_device(dev)
    if two_devices:
        dist.init_process_group("nccl", rank=rank, world_size=WORLD, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import preconditioned_stochastic_gradient_descent as prod
    params, U, V, d, probes = _cls_setup(dev)
    mine = slice(0, CLS_SPLIT) if rank == 0 else slice(CLS_SPLIT, None)
    own = [p.clone().requires_grad_(True) for p in params[mine]]
    lo = sum(p.numel() for p in params[:mine.start or 0])
    hi = lo + sum(p.numel() for p in own)
    gen = torch.Generator().manual_seed(606 + 13 * rank)                 # different seeds: rank 0's coins must win
    opt = psgd.UVd(own, rank_of_modification=CLS_R, lr_params=0.004, lr_preconditioner=0.05, generator=gen,
                   group=dist.group.WORLD, placement=("packed" if rank == 1 else None))      # (one rank with a placed state: same results)
    assert (opt._arena is not None) == (rank == 1)
    assert opt._num_params_global == sum(p.numel() for p in params) and tuple(opt._U.shape) == (hi - lo, CLS_R)
    opt._U.copy_(U[lo:hi]); opt._V.copy_(V[lo:hi]); opt._d.copy_(d[lo:hi])
    calls = {"all_gather_into_tensor": 0, "all_reduce": 0, "broadcast": 0}
    for name in calls:
        def wrap(fn, name=name):
            def counted(*a, **k):
                calls[name] += 1
                return fn(*a, **k)
            return counted
        setattr(dist, name, wrap(getattr(dist, name)))
    from psgd_tf_amd import sharded
    calls["exchanges"] = sharded.EXCHANGES["count"]
    saved, counts, last = {}, [], dict(calls)

    def snap(it):
        nonlocal last
        calls["exchanges"] = sharded.EXCHANGES["count"]
        counts.append([calls[k] - last[k] for k in ("exchanges", "all_gather_into_tensor", "all_reduce", "broadcast")])
        last = dict(calls)
        saved["U%d" % it], saved["V%d" % it], saved["d%d" % it] = (x.cpu().numpy().copy() for x in (opt._U, opt._V, opt._d))
        saved["p%d" % it] = torch.cat([p.detach().reshape(-1) for p in own]).cpu().numpy()
    _cls_run(opt, prod, [own], lambda it: [probes[it][mine]], lambda: _cls_loss(own, float(rank)), snap)
    np.savez(os.path.join(outdir, "cls%d.npz" % rank), counts=np.array(counts), **saved)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", [
    "gloo-one-device",
    pytest.param("rccl-two-devices", marks=pytest.mark.skipif(not _two_devices(), reason="needs two GPUs (rank k on cuda:k over RCCL)"))])
def test_sharded_uvd_class_equals_the_unsharded_class(hip_lib, transport):
    """UVd(..., group=pg) in two processes with the real stage kernels and real collectives against the unsharded class UVd on the
    concatenated parameters (psgd.py:692-764): clip on and off, a step that leaves the preconditioner alone, exact and
    finite-difference Hv.  1e-6 through the exact-Hv steps.  The finite-difference steps (:717-727) come last and get 1e-4: h is a
    difference of two fp32 gradients divided by 2^-11.5, so the last-bit differences between the two runs' states (their
    reductions are partitioned differently) come back multiplied by ~3e3 in h -- in ANY two runs that are not bit-identical."""
    import torch.multiprocessing as mp
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import preconditioned_stochastic_gradient_descent as prod
    outdir = tempfile.mkdtemp()
    port = 29300 + os.getpid() % 300
    two = transport == "rccl-two-devices"
    mp.start_processes(_cls_worker, args=(port, outdir, two), nprocs=WORLD, join=True, start_method="spawn")
    sh = [np.load(os.path.join(outdir, "cls%d.npz" % k)) for k in range(WORLD)]
    clip = [c != float("inf") for c in CLS_CLIP]
    # per step: 2 exchanges + the clip norm, no broadcast.  Over gloo they are torch.distributed calls; on an RCCL group they go
    # through the library's own communicator on the caller's stream (sharded._RcclDirect) and torch.distributed sees none
    for s in sh:
        assert s["counts"].tolist() == [[2, 0 if two else 2, 0 if two else int(c), 0] for c in clip], s["counts"]
    dev = torch.device("cuda:0")
    params, U, V, d, probes = _cls_setup(dev)
    allp = [p.clone().requires_grad_(True) for p in params]
    groups = [allp[:CLS_SPLIT], allp[CLS_SPLIT:]]
    opt = psgd.UVd(allp, rank_of_modification=CLS_R, lr_params=0.004, lr_preconditioner=0.05,
                   generator=torch.Generator().manual_seed(606))
    opt._U.copy_(U); opt._V.copy_(V); opt._d.copy_(d)
    keep = prod._randn_like
    try:
        def snap(it):
            tol = 1e-6 if all(CLS_EXACT[:it + 1]) else 1e-4
            got = {k: np.concatenate([s["%s%d" % (k, it)] for s in sh], 0) for k in ("U", "V", "d", "p")}
            want = {"U": opt._U, "V": opt._V, "d": opt._d, "p": torch.cat([p.detach().reshape(-1) for p in allp])}
            for k in got:
                assert rel_err(got[k], want[k].cpu().numpy()) < tol, (it, k, rel_err(got[k], want[k].cpu().numpy()))
        _cls_run(opt, prod, groups, lambda it: [probes[it][:CLS_SPLIT], probes[it][CLS_SPLIT:]],
                 lambda: _cls_loss(groups[0], 0.0) + _cls_loss(groups[1], 1.0), snap)
    finally:
        prod._randn_like = keep
