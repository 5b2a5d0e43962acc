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
SPLU_KEYS = ("L12", "l3", "U12", "u3")


def test_fixtures_present():
    assert len(UVD) == 4 and len(KRON) == 4 and len(SPLU) == 3


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
        if name.startswith("uvd_"):
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
