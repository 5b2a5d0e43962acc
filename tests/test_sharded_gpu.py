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
    # branch agreement path (rank 0 draws, broadcast) runs on the device
    gen = torch.Generator().manual_seed(3)
    sharded.update_precond_UVd_math_(b["U"], b["V"], b["d"], b["v"], b["h"], 0.01, TINY32, generator=gen)
    assert torch.isfinite(b["U"]).all() and torch.isfinite(b["d"]).all()
