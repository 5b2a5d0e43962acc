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


def _sparse_factor(shape, n, scale):
    """Initial factors as demo_usage_of_all_preconditioners.py:68-78 builds them."""
    if shape[0] == shape[1]:
        return scale * np.eye(n)                                   # dense
    if shape[0] == 2:
        return scale * np.stack([np.ones(n), np.zeros(n)], 0)      # normalization: [diagonal; last column]
    return scale * np.ones((1, n))                                 # scaling


def kron_sparse_case(name, shape_l, shape_r, scale_l, seed):
    """The six sparse dispatch formats (psgd.py:198-391) at the shapes of demo_usage_of_all_preconditioners.py:68-78
    (R = 5; I, J, K = 10, 20, 50).  Factors: the demo's initial ones advanced by two oracle updates so that the
    normalization factor's last column and the dense factor's upper triangle are populated."""
    M, N = shape_l[1], shape_r[1]
    assert orc.kron_format(shape_l, shape_r) == name.split("kron_fmt_")[1].rsplit("_", 1)[0]
    rng = np.random.default_rng(seed)
    Ql, Qr = _sparse_factor(shape_l, M, scale_l), _sparse_factor(shape_r, N, 1.0)
    for _ in range(2):
        x = rng.standard_normal((M, N))
        Ql, Qr = orc.update_precond_kron(Ql, Qr, x, x * np.exp(rng.uniform(-1, 1, (M, 1))) * np.exp(rng.uniform(-1, 1, (1, N))), 0.1)
    Ql, Qr = Ql.astype(np.float32), Qr.astype(np.float32)
    dX = rng.standard_normal((M, N)).astype(np.float32)
    dG = (np.diag(np.exp(rng.uniform(-1, 1, M))) @ dX @ np.diag(np.exp(rng.uniform(-1, 1, N)))).astype(np.float32)
    G = rng.standard_normal((M, N)).astype(np.float32)
    f = lambda a: a.astype(np.float64)
    Ql_new, Qr_new = orc.update_precond_kron(f(Ql), f(Qr), f(dX), f(dG), 0.1)     # step of the demo (:91)
    pre = orc.precond_grad_kron(f(Ql), f(Qr), f(G))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), Ql=Ql, Qr=Qr, dX=dX, dG=dG, G=G, step=0.1,
                        Ql_new=Ql_new, Qr_new=Qr_new, pre_grad=pre)


UVD_STEP_SHAPES = [(2, 30), (30, 30), (30,), (30, 1), (1,)]        # rnn_xor_UVd_preconditioner.py:28-31 -> 1021 parameters


def uvd_step_loss_terms(p, c, b, e):
    """The closure of the UVd.step fixture: loss = sum(c p^2 / 2 + b p + e p^4 / 4) over all parameters, so that
    grad = c p + b + e p^3 and H v = (c + 3 e p^2) v are known in closed form (inputs = explicit v, grads, Hv)."""
    loss = np.sum(0.5 * c * p * p + b * p + 0.25 * e * p ** 4)
    return loss, c * p + b + e * p ** 3, c + 3.0 * e * p * p


def uvd_step_case(name, seed, balance, update_U, max_norm):
    """One exact-Hv UVd.step (psgd.py:692-764) on the 1021-parameter layout: inputs are the parameters, the loss
    coefficients, the probe vector the step 'draws', the preconditioner state and the branch decisions."""
    rng = np.random.default_rng(seed)
    N, r = 1021, 10
    q = make_uvd_problem(N, r, seed=seed, uv_gain=2.0, d_spread=0.3)
    p = (0.3 * rng.standard_normal(N)).astype(np.float32)
    c = np.exp(rng.uniform(-2, 2, N)).astype(np.float32)
    b = (0.5 * rng.standard_normal(N)).astype(np.float32)
    e = np.exp(rng.uniform(-1, 1, N)).astype(np.float32)
    vs = rng.standard_normal(N).astype(np.float32)
    f = lambda a: a.astype(np.float64)
    loss, grad, hdiag = uvd_step_loss_terms(f(p), f(c), f(b), f(e))
    U, V, d = f(q["U"]), f(q["V"]), f(q["d"])
    unfl = lambda x: orc.uvd_unflatten(x, UVD_STEP_SHAPES)
    new = orc.uvd_step(unfl(f(p)), unfl(grad), unfl(hdiag * f(vs)), unfl(f(vs)), U, V, d, 0.05, 0.02, max_norm, TINY32,
                       balance=balance, update_U=update_U)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), p=p, c=c, b=b, e=e, vs=vs, U=q["U"], V=q["V"], d=q["d"],
                        rank=r, lr_params=0.05, lr_preconditioner=0.02, grad_clip_max_norm=max_norm, tiny=TINY32,
                        balance=balance, update_U=update_U, loss=loss, grad=grad, Hv=hdiag * f(vs),
                        p_new=orc.uvd_flatten(new, np.float64), U_new=U, V_new=V, d_new=d)


if __name__ == "__main__":
    uvd_case("uvd_n1021_r10_updU", 1021, 10, 1, 2.0, 0.3, False, True)     # rnn_xor model size (KAT-IDX)
    uvd_case("uvd_n1021_r10_updV_bal", 1021, 10, 2, 2.0, 0.3, True, False)
    uvd_case("uvd_n4096_r20_updU", 4096, 20, 3, 1.0, 0.0, False, True)     # reference init scales (psgd.py:687-690)
    uvd_case("uvd_n777_r1_updV", 777, 1, 4, 2.0, 0.2, False, False)
    kron_case("kron_lenet_w2_151x16", 151, 16, 5)                           # mnist_with_lenet5.py:13
    kron_case("kron_lenet_w5_85x10", 85, 10, 6)                             # mnist_with_lenet5.py:16
    kron_case("kron_wide_16x40", 16, 40, 7)                                 # M < N branch (psgd.py:189-190)
    kron_case("kron_1x1_3x3", 1, 3, 8)                                      # 1x1 dense factor (NMT demo :124)
    # sparse dispatch formats, demo_usage_of_all_preconditioners.py:68-70 (example 1) and :73-75 (example 2); R = 5
    kron_sparse_case("kron_fmt_dense_norm_5x10", (5, 5), (2, 10), 0.1, 21)
    kron_sparse_case("kron_fmt_scale_dense_5x20", (1, 5), (20, 20), 0.1, 22)
    kron_sparse_case("kron_fmt_scale_norm_5x50", (1, 5), (2, 50), 0.1, 23)
    kron_sparse_case("kron_fmt_norm_dense_5x10", (2, 5), (10, 10), 0.1, 24)
    kron_sparse_case("kron_fmt_dense_scale_5x20", (5, 5), (1, 20), 0.1, 25)
    kron_sparse_case("kron_fmt_norm_scale_5x50", (2, 5), (1, 50), 0.1, 26)
    uvd_step_case("uvdstep_n1021_r10_updU_clip", 31, False, True, 1.0)       # rnn_xor_UVd_preconditioner.py:33-36 (clip 1.0)
    uvd_step_case("uvdstep_n1021_r10_updV_bal_noclip", 32, True, False, np.inf)
    splu_case("splu_n400_r10_demo_init", 400, 10, 9, 0.1, demo_init=True)   # demo_usage_of_all_preconditioners.py:45-51,61
    splu_case("splu_n1021_r7", 1021, 7, 10, 0.01)                           # odd rank (alignment head path)
    splu_case("splu_n2048_r20", 2048, 20, 11, 0.1)
    print(sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
