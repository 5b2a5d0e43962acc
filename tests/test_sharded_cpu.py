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
        gen = torch.Generator().manual_seed(1234 + rank)          # different seeds: rank 0's draw must win
        sharded.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], 0.01, TINY, generator=gen,
                                         backend=be)
        out = sharded.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"], backend=be)
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), U=t["U"].numpy(), V=t["V"].numpy(), d=t["d"].numpy(),
                 out=out.numpy(), lo=lo, hi=hi)
        # fused update -> apply (one V pass less): both branches, continuing from that state
        outs_f = []
        for upd in (True, False):
            outs_f.append(sharded.update_precond_UVd_math_and_precond_grad(
                t["U"], t["V"], t["d"], t["v"], t["h"], t["g"], 0.01, TINY, balance=False, update_U=upd, backend=be))
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
