"""Generates the committed golden vectors in this directory.

The reference (Python on TensorFlow) cannot be imported in the build container, so these are
NOT outputs of the reference itself: they are outputs of the fp64 run of the CPU oracle
(oracle/psgd_oracle.py), which is pinned to the reference's source by tests/test_oracle_kat.py.
They freeze the oracle (any later edit that changes its results fails the golden tests) and give
the GPU tests inputs/outputs that do not depend on NumPy's RNG stream.

    python tests/golden/make_golden.py        (run from the repository root)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import psgd_oracle as orc          # noqa: E402
from tests.uvd_cases import make_uvd_problem   # noqa: E402
from tests.splu_cases import make_splu_problem  # noqa: E402

TINY32 = float(np.finfo(np.float32).tiny)


def uvd_case(name, N, r, seed, uv_gain, d_spread, balance, update_U):
    p = make_uvd_problem(N, r, seed=seed, uv_gain=uv_gain, d_spread=d_spread)
    if balance:
        p["U"] *= 6.0
    q = {k: v.astype(np.float64) for k, v in p.items()}
    pre0 = orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], 0.01, TINY32, balance=balance, update_U=update_U)
    pre1 = orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), U=p["U"], V=p["V"], d=p["d"], g=p["g"], v=p["v"], h=p["h"],
                        step=0.01, tiny=TINY32, balance=balance, update_U=update_U,
                        pre_grad_before=pre0, U_new=q["U"], V_new=q["V"], d_new=q["d"], pre_grad_after=pre1)


def kron_case(name, M, N, seed):
    rng = np.random.default_rng(seed)
    tri = lambda n: np.triu(rng.standard_normal((n, n)) * 0.05, 1) + np.diag(np.exp(0.3 * rng.standard_normal(n)))
    Ql, Qr = (tri(M) * 2.0).astype(np.float32), tri(N).astype(np.float32)
    dX = rng.standard_normal((M, N)).astype(np.float32)
    dG = (np.diag(np.exp(rng.uniform(-1, 1, M))) @ dX @ np.diag(np.exp(rng.uniform(-1, 1, N)))).astype(np.float32)
    G = rng.standard_normal((M, N)).astype(np.float32)
    f = lambda a: a.astype(np.float64)
    Ql_new, Qr_new = orc.update_precond_kron(f(Ql), f(Qr), f(dX), f(dG), 0.01)
    pre = orc.precond_grad_kron(f(Ql), f(Qr), f(G))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), Ql=Ql, Qr=Qr, dX=dX, dG=dG, G=G, step=0.01,
                        Ql_new=Ql_new, Qr_new=Qr_new, pre_grad=pre)


def splu_case(name, N, r, seed, step, demo_init=False):
    p = make_splu_problem(N, r, seed=seed, init_like_demo=demo_init)
    q = {k: v.astype(np.float64) for k, v in p.items()}
    pre = orc.precond_grad_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["g"]])[0]
    new = orc.update_precond_splu(q["L12"], q["l3"], q["U12"], q["u3"], [q["dx"]], [q["dg"]], step)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **p, step=step, pre_grad=pre, L12_new=new[0], l3_new=new[1],
                        U12_new=new[2], u3_new=new[3])


if __name__ == "__main__":
    uvd_case("uvd_n1021_r10_updU", 1021, 10, 1, 2.0, 0.3, False, True)     # rnn_xor model size (KAT-IDX)
    uvd_case("uvd_n1021_r10_updV_bal", 1021, 10, 2, 2.0, 0.3, True, False)
    uvd_case("uvd_n4096_r20_updU", 4096, 20, 3, 1.0, 0.0, False, True)     # reference init scales (psgd.py:687-690)
    uvd_case("uvd_n777_r1_updV", 777, 1, 4, 2.0, 0.2, False, False)
    kron_case("kron_lenet_w2_151x16", 151, 16, 5)                           # mnist_with_lenet5.py:13
    kron_case("kron_lenet_w5_85x10", 85, 10, 6)                             # mnist_with_lenet5.py:16
    kron_case("kron_wide_16x40", 16, 40, 7)                                 # M < N branch (psgd.py:189-190)
    kron_case("kron_1x1_3x3", 1, 3, 8)                                      # 1x1 dense factor (NMT demo :124)
    splu_case("splu_n400_r10_demo_init", 400, 10, 9, 0.1, demo_init=True)   # demo_usage_of_all_preconditioners.py:45-51,61
    splu_case("splu_n1021_r7", 1021, 7, 10, 0.01)                           # odd rank (alignment head path)
    splu_case("splu_n2048_r20", 2048, 20, 11, 0.1)
    print(sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
