"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the fp64 oracle).
CPU: the oracle still reproduces them (fp64, to 1e-13: BLAS summation order may differ between
hosts) and within rounding in fp32.
GPU: the HIP path matches them to the stated tolerances."""
import glob
import os

import numpy as np
import pytest

from oracle import psgd_oracle as orc
from tests.uvd_cases import rel_err

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UVD = sorted(glob.glob(os.path.join(HERE, "uvd_*.npz")))
KRON = sorted(glob.glob(os.path.join(HERE, "kron_*.npz")))
SPLU = sorted(glob.glob(os.path.join(HERE, "splu_*.npz")))
STEP = sorted(glob.glob(os.path.join(HERE, "uvdstep_*.npz")))
SPLU_KEYS = ("L12", "l3", "U12", "u3")
UVD_STEP_SHAPES = [(2, 30), (30, 30), (30,), (30, 1), (1,)]        # rnn_xor_UVd_preconditioner.py:28-31 -> 1021 parameters


def test_fixtures_present():
    assert len(UVD) == 4 and len(KRON) == 10 and len(SPLU) == 3 and len(STEP) == 2
    # the six sparse dispatch formats of psgd.py:86-104 are all there, at the demo's shapes
    fmts = sorted(os.path.basename(p)[len("kron_fmt_"):].rsplit("_", 1)[0] for p in KRON if "kron_fmt_" in p)
    assert fmts == ["dense_norm", "dense_scale", "norm_dense", "norm_scale", "scale_dense", "scale_norm"]
    for p in KRON:
        if "kron_fmt_" in p:
            z = np.load(p)
            assert "kron_fmt_" + orc.kron_format(z["Ql"].shape, z["Qr"].shape) in p


@pytest.mark.parametrize("path", UVD, ids=os.path.basename)
def test_oracle_reproduces_uvd_golden(path):
    z = np.load(path)
    q = {k: z[k].astype(np.float64) for k in ("U", "V", "d", "g", "v", "h")}
    assert rel_err(orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"]), z["pre_grad_before"]) < 1e-13
    orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], float(z["step"]), float(z["tiny"]),
                                 balance=bool(z["balance"]), update_U=bool(z["update_U"]))
    for k in ("U", "V", "d"):
        assert rel_err(q[k], z[k + "_new"]) < 1e-13, k
    # the fp32 run of the same op sequence stays within fp32 rounding of the golden
    p = {k: z[k].copy() for k in ("U", "V", "d", "g", "v", "h")}
    orc.update_precond_UVd_math_(p["U"], p["V"], p["d"], p["v"], p["h"], float(z["step"]), float(z["tiny"]),
                                 balance=bool(z["balance"]), update_U=bool(z["update_U"]))
    for k in ("U", "V", "d"):
        assert rel_err(p[k], z[k + "_new"]) < 1e-5
    assert rel_err(orc.precond_grad_UVd_math(p["U"], p["V"], p["d"], p["g"]), z["pre_grad_after"]) < 1e-5


@pytest.mark.parametrize("path", KRON, ids=os.path.basename)
def test_oracle_reproduces_kron_golden(path):
    z = np.load(path)
    f = lambda k: z[k].astype(np.float64)
    a, b = orc.update_precond_kron(f("Ql"), f("Qr"), f("dX"), f("dG"), float(z["step"]))
    assert rel_err(a, z["Ql_new"]) < 1e-13 and rel_err(b, z["Qr_new"]) < 1e-13
    assert rel_err(orc.precond_grad_kron(f("Ql"), f("Qr"), f("G")), z["pre_grad"]) < 1e-13


def _step_oracle(z, dt):
    f = lambda k: z[k].astype(dt)
    p, c, b, e, vs = f("p"), f("c"), f("b"), f("e"), f("vs")
    loss = np.sum(0.5 * c * p * p + b * p + 0.25 * e * p ** 4)
    grad, hv = c * p + b + e * p ** 3, (c + 3.0 * e * p * p) * vs
    U, V, d = f("U"), f("V"), f("d")
    unfl = lambda x: orc.uvd_unflatten(x, UVD_STEP_SHAPES)
    new = orc.uvd_step(unfl(p), unfl(grad), unfl(hv), unfl(vs), U, V, d, float(z["lr_params"]), float(z["lr_preconditioner"]),
                       float(z["grad_clip_max_norm"]), float(z["tiny"]), balance=bool(z["balance"]), update_U=bool(z["update_U"]))
    return loss, orc.uvd_flatten(new, dt), U, V, d


@pytest.mark.parametrize("path", STEP, ids=os.path.basename)
def test_oracle_reproduces_uvd_step_golden(path):
    z = np.load(path)
    loss, p_new, U, V, d = _step_oracle(z, np.float64)
    assert abs(loss - float(z["loss"])) < 1e-12 * abs(float(z["loss"]))
    for got, key in ((p_new, "p_new"), (U, "U_new"), (V, "V_new"), (d, "d_new")):
        assert rel_err(got, z[key]) < 1e-13, key
    assert rel_err(p_new, z["p"].astype(np.float64)) > 1e-3            # the step moved the parameters
    loss32, p32, U32, V32, d32 = _step_oracle(z, np.float32)           # fp32 run of the same op sequence
    for got, key in ((p32, "p_new"), (U32, "U_new"), (V32, "V_new"), (d32, "d_new")):
        assert rel_err(got, z[key]) < 1e-5, key


@pytest.mark.gpu
@pytest.mark.parametrize("path", STEP, ids=os.path.basename)
def test_hip_uvd_step_matches_golden(path, hip_lib, monkeypatch):
    """class UVd of the product on the fixture: same separable loss as a torch closure (autograd makes grads and Hv),
    the probe vectors and the three coin flips are the fixture's (injected where the step draws them)."""
    import torch
    import preconditioned_stochastic_gradient_descent as shim
    from psgd_tf_amd import preconditioned_stochastic_gradient_descent as mod
    z = np.load(path)
    dev = torch.device("cuda:0")

    def unfl(x):
        return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in orc.uvd_unflatten(x, UVD_STEP_SHAPES)]
    params = [t.requires_grad_(True) for t in unfl(z["p"])]
    cs, bs, es, vs = unfl(z["c"]), unfl(z["b"]), unfl(z["e"]), unfl(z["vs"])
    max_norm = float(z["grad_clip_max_norm"])
    opt = shim.UVd(params, rank_of_modification=int(z["rank"]), lr_params=float(z["lr_params"]),
                   lr_preconditioner=float(z["lr_preconditioner"]), grad_clip_max_norm=(None if np.isinf(max_norm) else max_norm))
    assert opt._param_sizes == [60, 900, 30, 30, 1] and opt._U.shape == (1021, 10)
    for t, k in ((opt._U, "U"), (opt._V, "V"), (opt._d, "d")):
        t.copy_(torch.from_numpy(z[k]).to(dev))
    coins = [True, bool(z["balance"]), bool(z["update_U"])]            # psgd.py:703, :562, :588 in the order they are drawn
    monkeypatch.setattr(mod, "_draw_branch", lambda p, gen: coins.pop(0))
    queue = list(vs)
    monkeypatch.setattr(torch, "randn_like", lambda t, **k: queue.pop(0))

    def closure():
        return sum(torch.sum(0.5 * c * p * p + b * p + 0.25 * e * p ** 4) for p, c, b, e in zip(params, cs, bs, es))
    loss = opt.step(closure)
    monkeypatch.undo()
    assert not coins and not queue
    assert abs(float(loss) - float(z["loss"])) < 1e-5 * abs(float(z["loss"]))
    p_new = np.concatenate([p.detach().reshape(-1).cpu().numpy() for p in params])
    assert rel_err(p_new, z["p_new"]) < 1e-5
    for t, k in ((opt._U, "U_new"), (opt._V, "V_new"), (opt._d, "d_new")):
        assert rel_err(t.cpu().numpy(), z[k]) < 1e-5, k


@pytest.mark.parametrize("path", SPLU, ids=os.path.basename)
def test_oracle_reproduces_splu_golden(path):
    z = np.load(path)
    f = lambda k: z[k].astype(np.float64)
    new = orc.update_precond_splu(f("L12"), f("l3"), f("U12"), f("u3"), [f("dx")], [f("dg")], float(z["step"]))
    for k, a in zip(SPLU_KEYS, new):
        assert rel_err(a, z[k + "_new"]) < 1e-13, k
    assert rel_err(orc.precond_grad_splu(f("L12"), f("l3"), f("U12"), f("u3"), [f("g")])[0], z["pre_grad"]) < 1e-13
    new32 = orc.update_precond_splu(z["L12"], z["l3"], z["U12"], z["u3"], [z["dx"]], [z["dg"]], float(z["step"]))
    for k, a in zip(SPLU_KEYS, new32):
        assert rel_err(a, z[k + "_new"]) < 1e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("path", SPLU, ids=os.path.basename)
def test_hip_matches_splu_golden(path, hip_lib):
    import torch
    import preconditioned_stochastic_gradient_descent as psgd
    z = np.load(path)
    c = lambda k: torch.from_numpy(z[k]).cuda()
    new = psgd.update_precond_splu(c("L12"), c("l3"), c("U12"), c("u3"), [c("dx")], [c("dg")], float(z["step"]))
    for k, a in zip(SPLU_KEYS, new):
        assert rel_err(a.cpu().numpy(), z[k + "_new"]) < 1e-5, k
    out = psgd.precond_grad_splu(c("L12"), c("l3"), c("U12"), c("u3"), [c("g")])[0]
    assert rel_err(out.cpu().numpy(), z["pre_grad"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("path", UVD, ids=os.path.basename)
def test_hip_matches_uvd_golden(path, hip_lib):
    import torch
    import preconditioned_stochastic_gradient_descent as psgd
    z = np.load(path)
    t = {k: torch.from_numpy(z[k]).cuda() for k in ("U", "V", "d", "g", "v", "h")}
    out0 = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
    assert rel_err(out0.cpu().numpy(), z["pre_grad_before"]) < 1e-5
    psgd.update_precond_UVd_math_(t["U"], t["V"], t["d"], t["v"], t["h"], float(z["step"]), float(z["tiny"]),
                                  balance=bool(z["balance"]), update_U=bool(z["update_U"]))
    for k in ("U", "V", "d"):
        assert rel_err(t[k].cpu().numpy(), z[k + "_new"]) < 1e-5, k
    out1 = psgd.precond_grad_UVd_math(t["U"], t["V"], t["d"], t["g"])
    assert rel_err(out1.cpu().numpy(), z["pre_grad_after"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("path", KRON, ids=os.path.basename)
def test_hip_matches_kron_golden(path, hip_lib):
    import torch
    import preconditioned_stochastic_gradient_descent as psgd
    z = np.load(path)
    c = lambda k: torch.from_numpy(z[k]).cuda()
    a, b = psgd.update_precond_kron(c("Ql"), c("Qr"), c("dX"), c("dG"), float(z["step"]))
    assert rel_err(a.cpu().numpy(), z["Ql_new"]) < 1e-5 and rel_err(b.cpu().numpy(), z["Qr_new"]) < 1e-5
    out = psgd.precond_grad_kron(c("Ql"), c("Qr"), c("G"))
    assert rel_err(out.cpu().numpy(), z["pre_grad"]) < 1e-5


# ----------------------------------------------------------------------------- reference-made fixtures (TensorFlow)
TF_DIR = os.path.join(HERE, "tf")
TF_FIXTURES = sorted(glob.glob(os.path.join(TF_DIR, "*.npz")))


def test_oracle_matches_reference_fixtures():
    """tests/golden/tf/*.npz are outputs of the REFERENCE ITSELF under TensorFlow for the inputs of the committed
    fixtures (tests/golden/make_golden_tf.py).  Present: the oracle must reproduce them (fp32 TensorFlow against the
    fp64 oracle: 1e-5, north_star's tolerance) and parity is pinned.  Absent (TensorFlow cannot be installed in the
    build container): this test says so -- PARITY UNPINNED -- instead of passing silently."""
    if not TF_FIXTURES:
        pytest.skip("PARITY UNPINNED: no reference-made fixtures in tests/golden/tf (TensorFlow unavailable; "
                    "run tests/golden/make_golden_tf.py where it is)")
    for path in TF_FIXTURES:
        name, t = os.path.basename(path), np.load(path)
        if name.startswith("dense_"):
            Q = orc.update_precond_dense(0.1 * np.eye(2), [np.array(1.0), np.array(0.0)], [np.array(802.0), np.array(400.0)], 0.2)
            assert rel_err(Q, t["Q_new"]) < 1e-5
            continue
        z = np.load(os.path.join(HERE, name))
        f = lambda k: z[k].astype(np.float64)
        if name.startswith("uvdstep_"):
            loss, p_new, U, V, d = _step_oracle(z, np.float64)
            assert abs(loss - float(t["loss"])) < 1e-5 * abs(loss), name
            for got, key in ((p_new, "p_new"), (U, "U_new"), (V, "V_new"), (d, "d_new")):
                assert rel_err(got, t[key]) < 1e-5, (name, key)
        elif name.startswith("uvd_"):
            q = {k: f(k) for k in ("U", "V", "d", "g", "v", "h")}
            assert rel_err(orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"]), t["pre_grad_before"]) < 1e-5, name
            orc.update_precond_UVd_math_(q["U"], q["V"], q["d"], q["v"], q["h"], float(z["step"]), float(z["tiny"]),
                                         balance=bool(z["balance"]), update_U=bool(z["update_U"]))
            for k in ("U", "V", "d"):
                assert rel_err(q[k], t[k + "_new"]) < 1e-5, (name, k)
            assert rel_err(orc.precond_grad_UVd_math(q["U"], q["V"], q["d"], q["g"]), t["pre_grad_after"]) < 1e-5, name
        elif name.startswith("kron_"):
            a, b = orc.update_precond_kron(f("Ql"), f("Qr"), f("dX"), f("dG"), float(z["step"]))
            assert rel_err(a, t["Ql_new"]) < 1e-5 and rel_err(b, t["Qr_new"]) < 1e-5, name
            assert rel_err(orc.precond_grad_kron(f("Ql"), f("Qr"), f("G")), t["pre_grad"]) < 1e-5, name
        else:
            new = orc.update_precond_splu(f("L12"), f("l3"), f("U12"), f("u3"), [f("dx")], [f("dg")], float(z["step"]))
            for k, a in zip(SPLU_KEYS, new):
                assert rel_err(a, t[k + "_new"]) < 1e-5, (name, k)
            assert rel_err(orc.precond_grad_splu(f("L12"), f("l3"), f("U12"), f("u3"), [f("g")])[0], t["pre_grad"]) < 1e-5
