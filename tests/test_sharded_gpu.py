"""GPU: the sharded driver with its product backend (HipStages) under a 1-rank nccl (RCCL) group
must reproduce the unsharded HIP path bit for bit (same kernels, same grid; the all-reduces are
identities at world size 1), and must match the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from oracle import psgd_oracle as orc
from tests.uvd_cases import TINY32, make_uvd_problem, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("N,r", [(50000, 20), (4099, 10)])
def test_sharded_world1_equals_unsharded(pg, hip_lib, N, r):
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import sharded
    p = make_uvd_problem(N, r, seed=8, uv_gain=2.0, d_spread=0.3)
    p["U"] *= 4.0
    a = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    b = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    q = {k: v.astype(np.float64) for k, v in p.items()}
    for bal, upd in ((True, True), (False, False)):
        psgd.update_precond_UVd_math_(a["U"], a["V"], a["d"], a["v"], a["h"], 0.01, TINY32, balance=bal, update_U=upd)
        sharded.update_precond_UVd_math_(b["U"], b["V"], b["d"], b["v"], b["h"], 0.01, TINY32, balance=bal, update_U=upd)
        orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=bal, update_U=upd)
    oa = psgd.precond_grad_UVd_math(a["U"], a["V"], a["d"], a["g"])
    ob = sharded.precond_grad_UVd_math(b["U"], b["V"], b["d"], b["g"])
    for k in ("U", "V", "d"):
        assert torch.equal(a[k], b[k]), k
        assert rel_err(b[k].cpu().numpy(), q[k]) < 1e-5
    assert torch.equal(oa, ob)
    assert rel_err(ob.cpu().numpy(), orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < 1e-5
    # fused update -> apply through the sharded driver == the unsharded fused call
    of_a = psgd.update_precond_UVd_math_and_precond_grad(a["U"], a["V"], a["d"], a["v"], a["h"], a["g"], 0.01, TINY32,
                                                         balance=False, update_U=True)
    of_b = sharded.update_precond_UVd_math_and_precond_grad(b["U"], b["V"], b["d"], b["v"], b["h"], b["g"], 0.01,
                                                            TINY32, balance=False, update_U=True)
    assert torch.equal(of_a, of_b) and torch.equal(a["U"], b["U"]) and torch.equal(a["d"], b["d"])   # same kernels
    # branch agreement path (generator state synchronised from rank 0 once, local draws afterwards)
    gen = torch.Generator().manual_seed(3)
    sharded.update_precond_UVd_math_(b["U"], b["V"], b["d"], b["v"], b["h"], 0.01, TINY32, generator=gen)
    assert torch.isfinite(b["U"]).all() and torch.isfinite(b["d"]).all()


def _allreduce_emulated(views, op):
    """What dist.all_reduce does across ranks, for the stage backends of several shards living in ONE process."""
    acc = views[0].clone()
    for v in views[1:]:
        acc = acc + v if op == "sum" else torch.maximum(acc, v)
    for v in views:
        v.copy_(acc)


def _gather_fold_emulated(bes, stage):
    """The product's exchange (sharded._exchange) across the stage backends of several shards living in ONE process:
    what all_gather_into_tensor delivers is the concatenation of the send regions in rank order; every "rank" then
    runs the product's fold kernel on it."""
    gathered = torch.cat([be.send(stage) for be in bes]).contiguous()
    for be in bes:
        be.fold(stage, gathered, len(bes))


@pytest.mark.parametrize("N,r,cuts", [(100003, 20, (0, 40000, 100003)), (5000, 7, (0, 1024, 1088, 5000))])
def test_uvd_hip_stages_on_real_shards(hip_lib, N, r, cuts):
    """Two / three row shards on one GPU, each with its own HipStages backend (the product's stage kernels on partial
    shards), the collectives emulated by summing / maxing the reduced workspace regions by hand: the concatenated
    result must equal the unsharded HIP call (to summation order) and the oracle."""
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import sharded
    p = make_uvd_problem(N, r, seed=12, uv_gain=2.0, d_spread=0.3)
    full = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    sh = [{k: torch.from_numpy(np.ascontiguousarray(v[a:b])).cuda() for k, v in p.items()} for a, b in zip(cuts, cuts[1:])]
    bes = [sharded.HipStages(torch.device("cuda:0"), s["U"].shape[0], r) for s in sh]
    q = {k: v.astype(np.float64) for k, v in p.items()}
    for upd in (True, False):
        psgd.update_precond_UVd_math_(full["U"], full["V"], full["d"], full["v"], full["h"], 0.01, TINY32, balance=False,
                                      update_U=upd)
        orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=upd)
        for be, s in zip(bes, sh):
            be.update_sweep1(s["U"], s["V"], s["d"], s["v"], s["h"])
        _allreduce_emulated([be.sums(11) for be in bes], "sum")
        for be, s in zip(bes, sh):
            be.update_sweep2(s["U"], s["V"], s["d"], s["v"], s["h"], 0.01, TINY32, upd)
        _allreduce_emulated([be.maxbuf(12) for be in bes], "max")
        for be, s in zip(bes, sh):
            be.update_sweep3(s["d"], 0.01, TINY32)
    for k in ("U", "V", "d"):
        got = torch.cat([s[k] for s in sh], 0)
        assert rel_err(got.cpu().numpy(), full[k].cpu().numpy()) < 1e-6, k
        assert rel_err(got.cpu().numpy(), q[k]) < 1e-5, k
    for be, s in zip(bes, sh):
        be.apply_sweep1(s["V"], s["d"], s["g"])
    _allreduce_emulated([be.sums(1) for be in bes], "sum")
    for be, s in zip(bes, sh):
        be.apply_sweep2(s["U"], s["d"], s["g"])
    _allreduce_emulated([be.sums(2) for be in bes], "sum")
    out = torch.cat([be.apply_sweep3(s["U"], s["V"], s["d"], s["g"]) for be, s in zip(bes, sh)], 0)
    assert rel_err(out.cpu().numpy(), orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < 1e-5


def test_splu_world1_and_real_shards(pg, hip_lib):
    """Sparse LU: (a) the sharded driver under a 1-rank RCCL group == the unsharded call bit for bit; (b) two real
    tail shards on one GPU with the collectives emulated by hand == the unsharded call and the oracle."""
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import sharded
    from tests.splu_cases import make_splu_problem
    N, r = 60007, 10
    p = make_splu_problem(N, r, seed=4)
    t = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    keys = ("L12", "l3", "U12", "u3")
    st = [t[k] for k in keys]
    want_new = psgd.update_precond_splu(*st, [t["dx"]], [t["dg"]], 0.1)
    want_out = psgd.precond_grad_splu(*want_new, [t["g"]])[0]
    # (a) world 1
    new1 = sharded.update_precond_splu(*st, t["dx"].reshape(-1), t["dg"].reshape(-1), 0.1)
    out1 = sharded.precond_grad_splu(*new1, t["g"].reshape(-1))
    assert all(torch.equal(a, b) for a, b in zip(new1, want_new)) and torch.equal(out1.reshape(-1), want_out.reshape(-1))
    # (b) two shards of the tail, corner replicated
    n2, cut = N - r, 23456
    def local(lo, hi):
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        d = {"L12": f(np.concatenate([p["L12"][:r], p["L12"][r + lo:r + hi]], 0)),
             "U12": f(np.concatenate([p["U12"][:, :r], p["U12"][:, r + lo:r + hi]], 1)),
             "l3": f(p["l3"][lo:hi]), "u3": f(p["u3"][lo:hi])}
        for k in ("dx", "dg", "g"):
            d[k] = f(np.concatenate([p[k][:r], p[k][r + lo:r + hi]], 0)).reshape(-1)
        return d
    sh = [local(0, cut), local(cut, n2)]
    bes = [sharded.HipSpluStages(torch.device("cuda:0"), s["L12"].shape[0], r) for s in sh]
    for be, s in zip(bes, sh):
        be.stage1(s["U12"], s["dg"])
    _allreduce_emulated([be.sums(1) for be in bes], "sum")
    for be, s in zip(bes, sh):
        be.update_stage2(s["L12"], s["l3"], s["U12"], s["u3"], s["dx"], s["dg"])
    _allreduce_emulated([be.sums(2) for be in bes], "sum")
    for be, s in zip(bes, sh):
        be.update_stage3(s["L12"], s["l3"], s["U12"], s["u3"], s["dx"], s["dg"])
    _gather_fold_emulated(bes, 3)                          # [r sums | 4 maxima] in one exchange
    news = [be.update_stage4(s["L12"], s["l3"], s["U12"], s["u3"], s["dx"], s["dg"], 0.1, float(psgd._tiny), True)
            for be, s in zip(bes, sh)]
    assert torch.equal(news[0][0][:r], news[1][0][:r]) and torch.equal(news[0][2][:, :r], news[1][2][:, :r])
    L12 = torch.cat([news[0][0], news[1][0][r:]], 0)
    U12 = torch.cat([news[0][2], news[1][2][:, r:]], 1)
    l3, u3 = torch.cat([news[0][1], news[1][1]], 0), torch.cat([news[0][3], news[1][3]], 0)
    q = {k: v.astype(np.float64) for k, v in p.items()}
    ref_new = orc.update_precond_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["dx"]], [q["dg"]], 0.1)
    for got, a, b in zip((L12, l3, U12, u3), want_new, ref_new):
        assert rel_err(got.cpu().numpy(), a.cpu().numpy()) < 1e-6
        assert rel_err(got.cpu().numpy(), b) < 1e-5
    # apply on the sharded new state
    for be, s, nw in zip(bes, sh, news):
        be.stage1(nw[2], s["g"])
    _allreduce_emulated([be.sums(1) for be in bes], "sum")
    for be, s, nw in zip(bes, sh, news):
        be.apply_stage2(nw[0], nw[1], nw[2], nw[3], s["g"])
    _allreduce_emulated([be.sums(2)[:r] for be in bes], "sum")
    outs = [be.apply_stage3(nw[0], nw[1], nw[2], nw[3]) for be, nw in zip(bes, news)]
    out = torch.cat([outs[0], outs[1][r:]], 0)
    assert torch.equal(outs[0][:r], outs[1][:r])
    assert rel_err(out.cpu().numpy(), orc.precond_grad_splu(*ref_new, [q["g"]])[0].reshape(-1)) < 1e-5


@pytest.mark.parametrize("N,r,cuts", [(100003, 20, (0, 40000, 100003)), (300001, 10, (0, 64, 150016, 300001))])
def test_uvd_fused_step_on_real_shards(hip_lib, N, r, cuts):
    """The choreography bench.py runs at --gpus N > 1 (sharded.update_precond_UVd_math_and_precond_grad): two
    exchanges -- Gram, [4r column sums | max] -- each an all-gather + the product's rank-order fold kernel, on real shards in
    one process with the gather emulated by concatenation, both branches."""
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import sharded
    p = make_uvd_problem(N, r, seed=13, uv_gain=2.0, d_spread=0.3)
    full = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    sh = [{k: torch.from_numpy(np.ascontiguousarray(v[a:b])).cuda() for k, v in p.items()} for a, b in zip(cuts, cuts[1:])]
    bes = [sharded.HipStages(torch.device("cuda:0"), s["U"].shape[0], r) for s in sh]
    q = {k: v.astype(np.float64) for k, v in p.items()}
    for upd in (True, False):
        want = psgd.update_precond_UVd_math_and_precond_grad(full["U"], full["V"], full["d"], full["v"], full["h"],
                                                             full["g"], 0.01, TINY32, balance=False, update_U=upd)
        orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=False, update_U=upd)
        for be, s in zip(bes, sh):
            be.update_sweep1(s["U"], s["V"], s["d"], s["v"], s["h"])
        _gather_fold_emulated(bes, 11)
        for be, s in zip(bes, sh):
            be.update_sweep2_fused(s["U"], s["V"], s["d"], s["v"], s["h"], s["g"], 0.01, TINY32, upd)
        _gather_fold_emulated(bes, 13)                     # [pU | pV | qU | qV | max] in one exchange
        assert all(torch.equal(bes[0].send(13), be.send(13)) for be in bes[1:])    # bit-identical on every "rank"
        for be in bes:
            be.fused_post(0.01, TINY32, upd)
        out = torch.cat([be.fused_final(s["U"], s["V"], s["d"], s["g"], 0.01, TINY32) for be, s in zip(bes, sh)], 0)
        assert rel_err(out.cpu().numpy(), want.cpu().numpy()) < 2e-6, upd
        assert rel_err(out.cpu().numpy(), orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])) < 1e-5, upd
        for k in ("U", "V", "d"):
            assert rel_err(torch.cat([s[k] for s in sh], 0).cpu().numpy(), full[k].cpu().numpy()) < 1e-6, (k, upd)
