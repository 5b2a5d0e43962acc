"""GPU, two PROCESSES on one device: the sharded drivers with the product's HIP stage kernels on each rank's row shard
and REAL collectives between the processes (gloo over device tensors: RCCL refuses two ranks on one GPU, and only one
GPU is available to the tests).  Complements tests/test_sharded_gpu.py (collectives emulated inside one process) and
tests/test_sharded_cpu.py (real collectives, NumPy stage kernels): here both are real.  Rank 0 compares the
concatenated shards with the unsharded HIP call and the oracle."""
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


def _worker(rank, port, outdir, N, r):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from psgd_tf_amd import sharded
    dev = torch.device("cuda:0")
    p = make_uvd_problem(N, r, seed=21, uv_gain=2.0, d_spread=0.3)
    lo, hi = sharded.shard_rows(N, rank, WORLD)
    t = {k: torch.from_numpy(np.ascontiguousarray(v[lo:hi])).to(dev) for k, v in p.items()}
    for upd in (True, False):
        sharded.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY32, balance=False, update_U=upd)
    out = sharded.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
    outf = sharded.update_precond_UVd_math_and_precond_grad(t["U"], t["V"], t["d"], t["v"], t["h"], t["g"], 0.01, TINY32,
                                                            balance=False, update_U=True)
    # sparse LU: the r x r corner replicated, the tail rows split between the ranks (tests/test_sharded_cpu.py layout)
    n, rr = N // 4, 7
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
    pre1 = sharded.precond_grad_splu(*new, loc["g"])
    np.savez(os.path.join(outdir, "r%d.npz" % rank), U=t["U"].cpu().numpy(), V=t["V"].cpu().numpy(), d=t["d"].cpu().numpy(),
             out=out.cpu().numpy(), outf=outf.cpu().numpy(), pre0=pre0.cpu().numpy(), pre1=pre1.cpu().numpy(),
             L12=new[0].cpu().numpy(), l3=new[1].cpu().numpy(), U12=new[2].cpu().numpy(), u3=new[3].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("N,r", [(200003, 20), (60003, 40)])      # r = 40: the wide-rank path (column chunks), sharded
def test_two_processes_real_kernels_real_collectives(hip_lib, N, r):
    import torch.multiprocessing as mp
    import preconditioned_stochastic_gradient_descent as psgd
    outdir = tempfile.mkdtemp()
    port = 29600 + os.getpid() % 300
    mp.start_processes(_worker, args=(port, outdir, N, r), nprocs=WORLD, join=True, start_method="spawn")
    sh = [np.load(os.path.join(outdir, "r%d.npz" % k)) for k in range(WORLD)]
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
    n, rr = N // 4, 7
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
