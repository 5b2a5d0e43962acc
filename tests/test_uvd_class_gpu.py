"""GPU: the UVd optimizer wrapper (psgd.py:630-764) -- state/index logic on the device, exact and
finite-difference Hessian-vector products through torch.autograd, clipping, the FD perturbation
undo -- on a small delayed-XOR-like RNN of the reference's size (rnn_xor_UVd_preconditioner.py:28-31:
1021 parameters in 5 tensors)."""
import math

import numpy as np
import pytest
import torch

from oracle import psgd_oracle as orc
from tests.uvd_cases import rel_err

pytestmark = pytest.mark.gpu


def _make_params(dev, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    shapes = [(2, 30), (30, 30), (30,), (30, 1), (1,)]
    return [(torch.randn(s, generator=g) * 0.3).to(dev).requires_grad_(True) for s in shapes]


def _xor_batch(dev, seed, batch=64, seq=8):
    rng = np.random.default_rng(seed)
    x = np.zeros((batch, seq, 2), dtype=np.float32)
    y = np.zeros((batch, 1), dtype=np.float32)
    for i in range(batch):
        x[i, :, 0] = rng.choice([-1.0, 1.0], seq)
        i1, i2 = int(rng.integers(0, 2)), int(rng.integers(2, seq))
        x[i, i1, 1] = x[i, i2, 1] = 1.0
        y[i] = -1.0 if x[i, i1, 0] == x[i, i2, 0] else 1.0
    return torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)


def _loss(params, x, y):
    Wx, Wh, bh, Wo, bo = params
    h = torch.zeros(x.shape[0], 30, device=x.device)
    for t in range(x.shape[1]):
        h = torch.tanh(x[:, t] @ Wx + h @ Wh + bh)
    return -torch.mean(torch.log(torch.sigmoid(y * (h @ Wo + bo))))     # rnn_xor_UVd_preconditioner.py:43-44


def test_init_state_and_index_logic(hip_lib):
    import preconditioned_stochastic_gradient_descent as psgd
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    params = _make_params(dev)
    opt = psgd.UVd(params, rank_of_modification=10, preconditioner_init_scale=0.5)
    assert opt._param_sizes == [60, 900, 30, 30, 1] and opt._param_cumsizes == [60, 960, 990, 1020, 1021]   # psgd.py:684-685
    assert opt._U.shape == (1021, 10) and opt._V.shape == (1021, 10) and opt._d.shape == (1021, 1)
    assert torch.all(opt._d == 0.5)                                                                       # psgd.py:690
    std = (1.0 / (1021 * 10)) ** 0.5                                                                     # psgd.py:687
    assert abs(float(opt._U.std()) / std - 1) < 0.05 and abs(float(opt._V.std()) / std - 1) < 0.05
    assert opt._tiny == float(np.finfo(np.float32).tiny) and opt._delta_param_scale == pytest.approx(2.0 ** -11.5)
    assert math.isinf(float(opt.grad_clip_max_norm))                                                     # psgd.py:675-676


def test_one_exact_step_matches_oracle(hip_lib):
    """One step with update probability 1 and exact Hv: replay the same (v, h, g) through the fp64 oracle."""
    import preconditioned_stochastic_gradient_descent as psgd
    dev = torch.device("cuda:0")
    params = _make_params(dev, seed=1)
    x, y = _xor_batch(dev, 1)
    gen = torch.Generator().manual_seed(5)
    opt = psgd.UVd(params, rank_of_modification=10, lr_params=0.05, lr_preconditioner=0.02, grad_clip_max_norm=1.0,
                   generator=gen)
    U0, V0, d0 = (t.cpu().numpy().astype(np.float64) for t in (opt._U, opt._V, opt._d))
    p0 = [p.detach().clone() for p in params]
    # reproduce the probe vectors the step will draw (torch.randn_like on the device uses the global generator)
    torch.manual_seed(123)
    vs = [torch.randn_like(p) for p in params]
    torch.manual_seed(123)
    closure = lambda: _loss(params, x, y)
    ret = opt.step(closure)
    assert torch.is_tensor(ret) and ret.dim() == 0
    # oracle replay
    grads = torch.autograd.grad(_loss(p0_req := [p.clone().requires_grad_(True) for p in p0], x, y), p0_req, create_graph=True)
    Hvs = torch.autograd.grad(grads, p0_req, vs)
    flat = lambda ts: np.concatenate([t.detach().reshape(-1).cpu().numpy().astype(np.float64) for t in ts])[:, None]
    v, h, g = flat(vs), flat(Hvs), flat(grads)
    gen2 = torch.Generator().manual_seed(5)
    assert torch.rand((), generator=gen2).item() < 1.0                      # the update_Q draw (psgd.py:703)
    bal = bool(torch.rand((), generator=gen2).item() < 0.01)
    upd = bool(torch.rand((), generator=gen2).item() < 0.5)
    orc.update_precond_UVd_math_(U0, V0, d0, v, h, 0.02, opt._tiny, balance=bal, update_U=upd)
    for got, ref in ((opt._U, U0), (opt._V, V0), (opt._d, d0)):
        assert rel_err(got.cpu().numpy(), ref) < 2e-5
    pre = orc.precond_grad_UVd_math(U0, V0, d0, g)
    lr = float(orc.uvd_clip_lr(pre, 0.05, 1.0, opt._tiny))                  # psgd.py:750-754
    new_flat = flat(p0) - lr * pre
    assert rel_err(flat(params), new_flat) < 2e-5
    # unflatten order: each tensor got its own slice (psgd.py:758-759)
    for p, ref in zip(params, orc.uvd_unflatten(new_flat[:, 0], [tuple(q.shape) for q in p0])):
        assert rel_err(p.detach().cpu().numpy(), ref) < 2e-5


def test_training_reduces_loss_exact_then_finite_difference(hip_lib):
    import preconditioned_stochastic_gradient_descent as psgd
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    params = _make_params(dev, seed=2)
    opt = psgd.UVd(params, rank_of_modification=10, lr_params=0.02, lr_preconditioner=0.02, grad_clip_max_norm=1.0,
                   preconditioner_update_probability=1.0, exact_hessian_vector_product=True)
    losses = []
    for it in range(300):
        x, y = _xor_batch(dev, 100 + it)
        if it == 150:
            opt.exact_hessian_vector_product.assign(False)                # rnn_xor_UVd_preconditioner.py:69
            opt.preconditioner_update_probability.assign(0.5)
        losses.append(float(opt.step(lambda: _loss(params, x, y)).detach()))
    assert all(math.isfinite(l) for l in losses)
    assert np.mean(losses[-20:]) < np.mean(losses[:20]) - 0.03        # learning, not diverging (XOR needs thousands of steps)
    assert all(torch.isfinite(t).all() for t in (opt._U, opt._V, opt._d))


def test_no_update_branch_leaves_preconditioner_alone(hip_lib):
    import preconditioned_stochastic_gradient_descent as psgd
    dev = torch.device("cuda:0")
    params = _make_params(dev, seed=3)
    x, y = _xor_batch(dev, 3)
    opt = psgd.UVd(params, preconditioner_update_probability=0.0)         # psgd.py:737-744
    U0, V0, d0 = opt._U.clone(), opt._V.clone(), opt._d.clone()
    opt.step(lambda: [_loss(params, x, y), "aux"])                         # closure may return a list (psgd.py:711)
    assert torch.equal(opt._U, U0) and torch.equal(opt._V, V0) and torch.equal(opt._d, d0)


@pytest.mark.parametrize("pdtype", [torch.bfloat16, torch.float16])
def test_half_precision_parameters(hip_lib, pdtype):
    """psgd.py:657-658: half-precision parameters are allowed.  Here: parameters, gradients and v, Hv in the half type,
    the preconditioner state and its arithmetic in fp32 (the kernels' type).  A convex quadratic goes down, the state
    stays fp32 and finite, _tiny / the FD scale follow the parameter type (:682-683), and one step equals the oracle's
    update + apply on the same widened (v, h, g)."""
    import preconditioned_stochastic_gradient_descent as psgd
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    A = (torch.randn(48, 48, device=dev) * 0.2)
    H = (A @ A.t() + 0.5 * torch.eye(48, device=dev)).to(pdtype)
    w = (torch.randn(48, 1, device=dev)).to(pdtype).requires_grad_(True)
    b = (torch.randn(7, device=dev)).to(pdtype).requires_grad_(True)
    closure = lambda: 0.5 * (w.t() @ H @ w).sum() + 0.5 * (b * b).sum()
    opt = psgd.UVd([w, b], rank_of_modification=5, lr_params=0.05, lr_preconditioner=0.05,
                   generator=torch.Generator().manual_seed(1))
    assert opt._U.dtype == torch.float32 and opt._d.dtype == torch.float32
    assert opt._tiny == float(torch.finfo(pdtype).tiny) and opt._delta_param_scale == float(torch.finfo(pdtype).eps) ** 0.5
    # one step replayed through the oracle
    U0, V0, d0 = (t.cpu().numpy().astype(np.float64) for t in (opt._U, opt._V, opt._d))
    seen = {}
    w0 = w.detach().clone()
    import psgd_tf_amd.preconditioned_stochastic_gradient_descent as impl
    impl_orig = impl.update_precond_UVd_math_and_precond_grad
    impl.update_precond_UVd_math_and_precond_grad = lambda *a, **k: _record(impl_orig, seen, *a, **k)
    try:
        loss0 = float(opt.step(closure).detach())
    finally:
        impl.update_precond_UVd_math_and_precond_grad = impl_orig
    assert seen["v"].dtype == torch.float32 and seen["g"].dtype == torch.float32
    f = lambda t: t.cpu().numpy().astype(np.float64)
    # the branch the step drew: U changed or V changed
    upd_U = not torch.equal(opt._U.cpu(), torch.from_numpy(U0.astype(np.float32)))
    orc.update_precond_UVd_math_(U0, V0, d0, f(seen["v"]), f(seen["h"]), 0.05, opt._tiny, balance=False, update_U=upd_U)
    want = orc.precond_grad_UVd_math(U0, V0, d0, f(seen["g"]))
    assert rel_err(f(seen["out"]), want) < 1e-5
    assert rel_err(f(opt._d), d0) < 1e-5
    # parameters moved by lr * pre_grad; the subtraction rounds to the half type (ulp(1) / step = 8 % per element in
    # bf16, 1 % in fp16), so this only has to catch a wrong slice or sign
    step_w = (w0.float() - w.detach().float()).cpu().numpy()
    assert rel_err(step_w, 0.05 * want[:48].reshape(48, 1)) < (0.15 if pdtype == torch.bfloat16 else 0.03)
    for _ in range(60):
        loss = float(opt.step(closure).detach())
    assert loss < 0.5 * loss0 and math.isfinite(loss)
    assert torch.isfinite(opt._U).all() and torch.isfinite(opt._d).all() and w.dtype == pdtype


def _record(fn, seen, U, V, d, v, h, g, **kw):
    out = fn(U, V, d, v, h, g, **kw)
    seen.update(v=v.clone(), h=h.clone(), g=g.clone(), out=out.clone())
    return out


@pytest.mark.parametrize("pdtype", [torch.bfloat16, torch.float16])
def test_state_in_the_parameters_dtype(hip_lib, pdtype):
    """state_dtype="param" (extension): U, V, d are STORED in the parameters' type as in the reference (psgd.py:688-690), widened
    for the fp32 kernels and rounded back once per step; the default keeps an fp32 state.  Runs, stays finite, the loss goes down,
    and the stored state equals the rounded fp32 state of a twin optimizer after one step from the same seed."""
    import preconditioned_stochastic_gradient_descent as psgd
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    A = torch.randn(200, 200, device=dev) * 0.1
    H = (A @ A.t() + 0.3 * torch.eye(200, device=dev))

    def make(state_dtype):
        torch.manual_seed(3)
        w = (torch.randn(200, 1, device=dev) * 0.5).to(pdtype).requires_grad_(True)
        opt = psgd.UVd([w], rank_of_modification=10, lr_params=0.05, lr_preconditioner=0.05, generator=torch.Generator().manual_seed(2),
                       state_dtype=state_dtype)
        return w, opt, (lambda: 0.5 * (w.float().t() @ H @ w.float()).sum())
    w, opt, closure = make("param")
    assert opt._U.dtype == pdtype and opt._V.dtype == pdtype and opt._d.dtype == pdtype
    w2, opt2, closure2 = make(None)
    assert opt2._U.dtype == torch.float32
    opt2._U.copy_(opt._U.float()); opt2._V.copy_(opt._V.float()); opt2._d.copy_(opt._d.float())   # the same (rounded) starting state
    torch.manual_seed(5)
    l0 = float(opt.step(closure).detach())
    torch.manual_seed(5)
    opt2.step(closure2)
    assert torch.equal(opt._U, opt2._U.to(pdtype)) and torch.equal(opt._d, opt2._d.to(pdtype))
    for _ in range(60):
        l = float(opt.step(closure).detach())
    assert torch.isfinite(opt._U.float()).all() and torch.isfinite(w.float()).all() and l < 0.5 * l0
    with pytest.raises(TypeError):
        psgd.UVd([w], state_dtype=torch.float64)
