"""MI355X-native PSGD preconditioner engine -- host-side mirror of the reference module.

Same module name, function names, positional order and defaults as the reference
``preconditioned_stochastic_gradient_descent.py`` ("psgd.py" in the citations), so the
demo drivers' call patterns (``import preconditioned_stochastic_gradient_descent as psgd``,
hello_psgd.py:5) carry over with torch tensors in place of tf tensors:

    update_precond_dense(Q, dxs, dgs, step=0.01) -> Q                  psgd.py:26
    precond_grad_dense(Q, grads) -> list                               psgd.py:45
    update_precond_kron(Ql, Qr, dX, dG, step=0.01) -> (Ql, Qr)         psgd.py:72
    precond_grad_kron(Ql, Qr, Grad) -> Tensor                          psgd.py:116
    update_precond_splu(L12, l3, U12, u3, dxs, dgs, step=0.01) -> 4    psgd.py:396
    precond_grad_splu(L12, l3, U12, u3, grads) -> list                 psgd.py:483
    IpUVtmatvec(U, V, x)                                               psgd.py:540
    update_precond_UVd_math_(U, V, d, v, h, step, tiny) -> None        psgd.py:554  (in place)
    precond_grad_UVd_math(U, V, d, g) -> Tensor                        psgd.py:619
    class UVd(...).step(closure)                                       psgd.py:630

The UVd, sparse-LU and Kron arithmetic runs in hand-written HIP kernels behind the
C ABI of include/psgd_hip.h (bound in _lib.py).  Tensors must be fp32 and resident on a ROCm
device; anything else raises -- there is no CPU fallback for the hot path.  Strided views are
accepted (copied on the way in, in-place state written back on the way out); the kernels work on
contiguous memory.  The dense preconditioner (psgd.py:26-63) is host-side plumbing on torch ops
(2x2 matrices, never a kernel target).

The reference draws its two branch decisions (psgd.py:562, :588) from TensorFlow's
global RNG; here they come from a torch.Generator (module default, or ``generator=``)
or are fixed through the keyword-only ``balance=`` / ``update_U=`` arguments.
"""
import ctypes
import os
import math

import torch

from . import _lib
from . import kron as _kron
from . import uvd_wide as _wide
from . import splu_wide as _splu_wide

dtype = torch.float32                                  # psgd.py:20
_tiny = torch.finfo(torch.float32).tiny                # psgd.py:22 (smallest normal fp32)

_branch_rng = torch.Generator(device="cpu")
_branch_rng.manual_seed(0x50534744)


def manual_seed(seed):
    """Seed the generator behind the branch draws of update_precond_UVd_math_ and UVd.step."""
    _branch_rng.manual_seed(int(seed))


# --------------------------------------------------------------------------- helpers
_ws_cache = _lib.WorkspaceCache()


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def _require_hip(name, *tensors):
    for t in tensors:
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s: expected torch tensors, got %r" % (name, type(t)))
        if not t.is_cuda:
            raise _lib.PsgdHipError("%s runs on the HIP device only (tensor is on %s); no CPU fallback" % (name, t.device))
        if t.dtype != torch.float32:
            raise TypeError("%s: fp32 tensors required, got %s" % (name, t.dtype))
        if not t.is_contiguous():
            raise ValueError("%s: contiguous tensors required" % name)
    dev = tensors[0].device
    for t in tensors:
        if t.device != dev:
            raise ValueError("%s: all tensors must be on one device" % name)
    return dev


def _c(t):
    """Read-only operand as the kernels need it (contiguous): a strided view is copied, as TensorFlow's ops would
    materialise it -- the reference has no notion of a non-contiguous tensor (SURVEY 8b's `ldU, ldV` are not in the ABI)."""
    return t if (not isinstance(t, torch.Tensor)) or t.is_contiguous() else t.contiguous()


class _InPlace:
    """State tensors a call updates in place (U, V, d): contiguous ones are used as they are, strided views are worked on
    as contiguous copies and written back when the call returns."""

    def __init__(self, *tensors):
        self.orig = tensors
        self.work = tuple(_c(t) for t in tensors)

    def writeback(self):
        for o, w in zip(self.orig, self.work):
            if w is not o:
                o.copy_(w)


def uvd_workspace(device, N, r):
    """Cached device workspace for a shard of N rows at rank r (see psgd_uvd_workspace_bytes).  Keyed by the current
    stream as well: a workspace carries the reduced vectors between the sweeps of one call, so calls issued on two
    streams must not share one (the C ABI itself is safe on several streams as long as each has its own workspace)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(N), int(r),
           torch.cuda.current_stream(device).cuda_stream)

    def make():
        nbytes = _lib.load().psgd_uvd_workspace_bytes(N, r)
        if nbytes < 0:
            _lib.check(int(nbytes), "psgd_uvd_workspace_bytes")
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return _ws_cache.get(key, make)


def _uvd_shapes(name, U, V, *cols):
    if U.dim() != 2 or V.shape != U.shape:
        raise ValueError("%s: U and V must both be [N, r]" % name)
    N, r = U.shape
    for c in cols:
        if c.numel() != N or (c.dim() == 2 and c.shape[1] != 1) or c.dim() > 2:
            raise ValueError("%s: column vectors must be [N] or [N, 1] with N = %d" % (name, N))
    return N, r


# --------------------------------------------------------------------------- dense (plumbing)
def update_precond_dense(Q, dxs, dgs, step=0.01):
    """psgd.py:26-42.  Host-side plumbing on torch ops (config 1, hello_psgd: 2x2 on CPU)."""
    dx = torch.cat([torch.reshape(x, [-1, 1]) for x in dxs], 0)
    dg = torch.cat([torch.reshape(g, [-1, 1]) for g in dgs], 0)
    a = Q @ dg
    b = torch.linalg.solve_triangular(Q.t(), dx, upper=False)      # Q^T b = dx  (:39, adjoint=True)
    grad = torch.triu(a @ a.t() - b @ b.t())
    step0 = step / (torch.max(torch.abs(grad)) + torch.finfo(Q.dtype).tiny)
    return Q - (step0 * grad) @ Q


def precond_grad_dense(Q, grads):
    """psgd.py:45-63: list in, list out with the original shapes."""
    cols = [torch.reshape(g, [-1, 1]) for g in grads]
    lens = [c.shape[0] for c in cols]
    grad = torch.cat(cols, 0)
    pre_grad = Q.t() @ (Q @ grad)
    pre_grads, idx = [], 0
    for g, n in zip(grads, lens):
        pre_grads.append(torch.reshape(pre_grad[idx:idx + n], g.shape))
        idx = idx + n
    return pre_grads


# --------------------------------------------------------------------------- Kron
def update_precond_kron(Ql, Qr, dX, dG, step=0.01):
    """psgd.py:72-110 (shape dispatch of SURVEY Appendix B); returns (Ql_new, Qr_new)."""
    return _kron.update_precond_kron(Ql, Qr, dX, dG, step)


def precond_grad_kron(Ql, Qr, Grad):
    """psgd.py:116-152; returns the preconditioned gradient, same shape as Grad."""
    return _kron.precond_grad_kron(Ql, Qr, Grad)


def update_precond_kron_batched(Qls, Qrs, dXs, dGs, step=0.01):
    """Extension: the list comprehension of mnist_with_lenet5.py:51 as one batched call."""
    return _kron.update_precond_kron_batched(Qls, Qrs, dXs, dGs, step)


def precond_grad_kron_batched(Qls, Qrs, Grads):
    """Extension: the list comprehension of mnist_with_lenet5.py:53 as one batched call."""
    return _kron.precond_grad_kron_batched(Qls, Qrs, Grads)


# --------------------------------------------------------------------------- sparse LU
def _splu_workspace(device, N, r):
    key = ("splu", device.index if device.index is not None else torch.cuda.current_device(), int(N), int(r),
           torch.cuda.current_stream(device).cuda_stream)

    def make():
        nbytes = int(_lib.load().psgd_splu_workspace_bytes(N, r))
        if nbytes <= 0:
            raise _lib.PsgdHipError("psgd_splu_workspace_bytes: unsupported shape N=%d r=%d (1 <= r <= 32, N >= r)" % (N, r))
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    return _ws_cache.get(key, make)


def _splu_shapes(name, L12, l3, U12, u3):
    if L12.dim() != 2 or U12.dim() != 2 or U12.shape != (L12.shape[1], L12.shape[0]):
        raise ValueError("%s: L12 must be [N, r] and U12 [r, N]" % name)
    N, r = L12.shape
    for c in (l3, u3):
        if c.numel() != N - r or c.dim() > 2 or (c.dim() == 2 and c.shape[1] != 1):
            raise ValueError("%s: l3 and u3 must be [N - r, 1] with N - r = %d" % (name, N - r))
    return N, r


def _tall(name, xs, N):
    """psgd.py:426-427 / :495-497: the list as one tall column vector (device-side cat; plumbing)."""
    flat = torch.cat([torch.reshape(x, [-1]) for x in xs], 0) if len(xs) != 1 else torch.reshape(xs[0], [-1])
    if flat.numel() != N:
        raise ValueError("%s: the list holds %d elements, the preconditioner is for %d" % (name, flat.numel(), N))
    return flat.contiguous()


def _splu_chunked(r):
    """ranks above PSGD_SPLU_MAX_RANK (64) run on column chunks (splu_wide.py); PSGD_SPLU_CHUNKS=1 sends ranks 33 .. 64 there too
    (the route of rounds 3-4: A/B runs of tools/r05_wide_rank.py)."""
    return r > _lib.SPLU_MAX_RANK or (r > _lib.UVD_MAX_RANK and os.environ.get("PSGD_SPLU_CHUNKS") == "1")


def update_precond_splu(L12, l3, U12, u3, dxs, dgs, step=0.01):
    """psgd.py:396-480: returns (L12_new, l3_new, U12_new, u3_new); inputs are not modified."""
    L12, l3, U12, u3 = _c(L12), _c(l3), _c(U12), _c(u3)
    dev = _require_hip("update_precond_splu", L12, l3, U12, u3)
    N, r = _splu_shapes("update_precond_splu", L12, l3, U12, u3)
    dx, dg = _tall("update_precond_splu", dxs, N), _tall("update_precond_splu", dgs, N)
    _require_hip("update_precond_splu", dx, dg, L12)
    if _splu_chunked(r):                           # wide rank: column chunks of L2 and U2' (splu_wide.py)
        return _splu_wide.update(L12, l3, U12, u3, dx, dg, float(step), float(_tiny), uvd_workspace)
    out = [torch.empty_like(t) for t in (L12, l3, U12, u3)]
    ws = _splu_workspace(dev, N, r)
    rc = _lib.load().psgd_splu_update_f32(L12.data_ptr(), l3.data_ptr(), U12.data_ptr(), u3.data_ptr(), dx.data_ptr(),
                                          dg.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                          out[3].data_ptr(), N, r, float(step), float(_tiny), ws.data_ptr(), ws.numel(),
                                          _stream_ptr(dev))
    _lib.check(rc, "psgd_splu_update_f32")
    return tuple(out)


def precond_grad_splu(L12, l3, U12, u3, grads):
    """psgd.py:483-524: list of gradients in, list of preconditioned gradients (same shapes) out."""
    L12, l3, U12, u3 = _c(L12), _c(l3), _c(U12), _c(u3)
    dev = _require_hip("precond_grad_splu", L12, l3, U12, u3)
    N, r = _splu_shapes("precond_grad_splu", L12, l3, U12, u3)
    g = _tall("precond_grad_splu", grads, N)
    _require_hip("precond_grad_splu", g, L12)
    if _splu_chunked(r):
        out = _splu_wide.precond_grad(L12, l3, U12, u3, g, uvd_workspace)
        pre_grads, idx = [], 0
        for x in grads:
            n = x.numel()
            pre_grads.append(torch.reshape(out[idx:idx + n], x.shape))
            idx += n
        return pre_grads
    out = torch.empty_like(g)
    ws = _splu_workspace(dev, N, r)
    rc = _lib.load().psgd_splu_apply_f32(L12.data_ptr(), l3.data_ptr(), U12.data_ptr(), u3.data_ptr(), g.data_ptr(),
                                         out.data_ptr(), N, r, ws.data_ptr(), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "psgd_splu_apply_f32")
    pre_grads, idx = [], 0                                                       # :518-522
    for x in grads:
        n = x.numel()
        pre_grads.append(torch.reshape(out[idx:idx + n], x.shape))
        idx += n
    return pre_grads


# --------------------------------------------------------------------------- UVd math
def IpUVtmatvec(U, V, x):
    """psgd.py:540-544: (I + U V') x for a column vector x ([N] or [N,1]) or [N,k] matrix."""
    U, V = _c(U), _c(V)
    if not (isinstance(x, torch.Tensor) and x.dim() == 2 and x.shape[1] > 1):
        x = _c(x)
    if U.dim() == 2 and U.shape[1] > _lib.UVD_MAX_RANK:
        _require_hip("IpUVtmatvec", U, V, x)
        return _wide.ipuvt_matvec(U, V, x, uvd_workspace)
    if x.dim() == 2 and x.shape[1] > 1:
        # a matrix x: its columns go through as contiguous vectors (two small transposes, N k floats each), U and V are
        # swept once per group of four columns
        dev = _require_hip("IpUVtmatvec", U, V)
        if not x.is_cuda or x.dtype != torch.float32:
            _require_hip("IpUVtmatvec", x)
        N, r = _uvd_shapes("IpUVtmatvec", U, V)
        if x.shape[0] != N:
            raise ValueError("IpUVtmatvec: x must have N = %d rows" % N)
        k = x.shape[1]
        xt = x.t().contiguous()
        ot = torch.empty_like(xt)
        ws = uvd_workspace(dev, N, r)
        xs = (ctypes.c_void_p * k)(*[xt[j].data_ptr() for j in range(k)])
        os_ = (ctypes.c_void_p * k)(*[ot[j].data_ptr() for j in range(k)])
        rc = _lib.load().psgd_uvd_ipuvt_matvec_cols_f32(U.data_ptr(), V.data_ptr(), xs, os_, k, N, r, ws.data_ptr(),
                                                         ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "psgd_uvd_ipuvt_matvec_cols_f32")
        return ot.t().contiguous()
    dev = _require_hip("IpUVtmatvec", U, V, x)
    N, r = _uvd_shapes("IpUVtmatvec", U, V, x)
    out = torch.empty_like(x)
    ws = uvd_workspace(dev, N, r)
    rc = _lib.load().psgd_uvd_ipuvt_matvec_f32(U.data_ptr(), V.data_ptr(), x.data_ptr(), out.data_ptr(), N, r,
                                                ws.data_ptr(), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "psgd_uvd_ipuvt_matvec_f32")
    return out


def _cols_of(name, x, N):
    """A matrix operand [N, k] as k contiguous columns: its transpose, materialised once ([k, N] row-major; a transposed view of a
    [k, N] tensor is used as it is)."""
    if x.shape[0] != N:
        raise ValueError("%s: the matrix must have N = %d rows" % (name, N))
    xt = x.t()
    return xt if xt.is_contiguous() else xt.contiguous()


def precond_grad_UVd_math(U, V, d, g, *, out=None):
    """psgd.py:619-627: d .* (I + V U')(I + U V')(d .* g); returns a new tensor shaped like g.
    g is a column vector ([N] or [N, 1]) or, as the reference's docstring allows (:623), a matrix [N, k]: d broadcasts over
    the columns and U, V are swept once per group of four columns (psgd_uvd_apply_cols_f32).
    out (extension; column-vector g, r <= 32): a contiguous fp32 tensor shaped like g to write the result to (placement.UVdArena.out)."""
    U, V, d = _c(U), _c(V), _c(d)
    if isinstance(g, torch.Tensor) and g.dim() == 2 and g.shape[1] > 1:
        dev = _require_hip("precond_grad_UVd_math", U, V, d)
        if not g.is_cuda or g.dtype != torch.float32 or g.device != dev:
            _require_hip("precond_grad_UVd_math", U, g)
        N, r = _uvd_shapes("precond_grad_UVd_math", U, V, d)
        gt = _cols_of("precond_grad_UVd_math", g, N)
        k = gt.shape[0]
        if r > _lib.UVD_MAX_RANK:
            return _wide.precond_grad(U, V, d, [gt[j] for j in range(k)], uvd_workspace).t().contiguous()
        ot = torch.empty_like(gt)
        ws = uvd_workspace(dev, N, r)
        gs = (ctypes.c_void_p * k)(*[gt[j].data_ptr() for j in range(k)])
        os_ = (ctypes.c_void_p * k)(*[ot[j].data_ptr() for j in range(k)])
        rc = _lib.load().psgd_uvd_apply_cols_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), gs, os_, k, N, r, ws.data_ptr(),
                                                  ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "psgd_uvd_apply_cols_f32")
        return ot.t().contiguous()
    g = _c(g)
    dev = _require_hip("precond_grad_UVd_math", U, V, d, g)
    N, r = _uvd_shapes("precond_grad_UVd_math", U, V, d, g)
    if r > _lib.UVD_MAX_RANK:                      # wide rank: column chunks through the same kernels (uvd_wide.py)
        return _wide.precond_grad(U, V, d, g, uvd_workspace)
    if out is None:
        out = torch.empty_like(g)
    else:
        _require_hip("precond_grad_UVd_math", out, U)
        if out.shape != g.shape:
            raise ValueError("precond_grad_UVd_math: out must be shaped like g")
    ws = uvd_workspace(dev, N, r)
    rc = _lib.load().psgd_uvd_apply_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), g.data_ptr(), out.data_ptr(),
                                         N, r, ws.data_ptr(), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "psgd_uvd_apply_f32")
    return out


def _draw_branch(p, generator):
    gen = generator if generator is not None else _branch_rng
    return bool(torch.rand((), generator=gen).item() < p)


def update_precond_UVd_math_(U, V, d, v, h, step, tiny, *, balance=None, update_U=None, generator=None):
    """psgd.py:554-617.  Updates U or V, and d, IN PLACE; returns None.

    balance / update_U fix the two random branches of the reference (:562 p=0.01, :588 p=0.5);
    left at None they are drawn from `generator` (a CPU torch.Generator; module default otherwise),
    in the reference's order."""
    state = _InPlace(U, V, d)
    (U, V, d), v, h = state.work, _c(v), _c(h)
    dev = _require_hip("update_precond_UVd_math_", U, V, d, v, h)
    N, r = _uvd_shapes("update_precond_UVd_math_", U, V, d, v, h)
    if balance is None:
        balance = _draw_branch(0.01, generator)
    if update_U is None:
        update_U = _draw_branch(0.5, generator)
    if r > _lib.UVD_MAX_RANK:
        _wide.update(U, V, d, v, h, float(step), float(tiny), bool(balance), bool(update_U), uvd_workspace)
        state.writeback()
        return None
    ws = uvd_workspace(dev, N, r)
    rc = _lib.load().psgd_uvd_update_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(), h.data_ptr(), N, r,
                                          float(step), float(tiny), int(bool(balance)), int(bool(update_U)),
                                          ws.data_ptr(), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "psgd_uvd_update_f32")
    state.writeback()
    return None


def update_precond_UVd_math_and_precond_grad(U, V, d, v, h, g, step, tiny, *, balance=None, update_U=None,
                                             generator=None, out=None):
    """Extension: update_precond_UVd_math_(U, V, d, v, h, step, tiny) followed by
    precond_grad_UVd_math(U, V, d, g) on the updated state -- the UVd.step pattern (psgd.py:732 -> :748) --
    as one fused call that saves a pass over V.  U or V, and d, are updated in place; returns the
    preconditioned gradient.  out (optional): a contiguous fp32 tensor shaped like g to write it to (placement.UVdArena.out:
    where the output stream lives is worth 4 % of the last sweep); ranks above 32 ignore it."""
    state = _InPlace(U, V, d)
    (U, V, d), v, h, g = state.work, _c(v), _c(h), _c(g)
    dev = _require_hip("update_precond_UVd_math_and_precond_grad", U, V, d, v, h, g)
    N, r = _uvd_shapes("update_precond_UVd_math_and_precond_grad", U, V, d, v, h, g)
    if balance is None:
        balance = _draw_branch(0.01, generator)
    if update_U is None:
        update_U = _draw_branch(0.5, generator)
    if r > _lib.UVD_MAX_RANK:                      # ranks 33 .. 64: the fused sequence of uvd_wide.update_apply; above: update, then apply
        out = _wide.update_apply(U, V, d, v, h, g, float(step), float(tiny), bool(balance), bool(update_U), uvd_workspace)
        state.writeback()
        return out
    if out is None:
        out = torch.empty_like(g)
    else:
        _require_hip("update_precond_UVd_math_and_precond_grad", out, U)
        if out.shape != g.shape:
            raise ValueError("update_precond_UVd_math_and_precond_grad: out must be shaped like g")
    ws = uvd_workspace(dev, N, r)
    rc = _lib.load().psgd_uvd_update_apply_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(), h.data_ptr(),
                                                g.data_ptr(), out.data_ptr(), N, r, float(step), float(tiny),
                                                int(bool(balance)), int(bool(update_U)), ws.data_ptr(), ws.numel(),
                                                _stream_ptr(dev))
    _lib.check(rc, "psgd_uvd_update_apply_f32")
    state.writeback()
    return out


# --------------------------------------------------------------------------- UVd optimizer wrapper
class _Hyper:
    """Stand-in for the non-trainable tf.Variable hyper-parameters of psgd.py:673-680:
    change them with .assign(value), read them with float()/bool()."""

    def __init__(self, value):
        self.value = value

    def assign(self, value):
        self.value = value.value if isinstance(value, _Hyper) else value
        return self

    def numpy(self):
        return self.value

    def __float__(self):
        return float(self.value)

    def __bool__(self):
        return bool(self.value)

    def __repr__(self):
        return "_Hyper(%r)" % (self.value,)


def _randn_like(p):
    """The draw of psgd.py:713 / :721 (one place, so that a test can supply the global vector's slices)."""
    return torch.randn_like(p)


def uvd_param_index(params):
    """psgd.py:684-686: sizes and cumulative sizes of the parameters, in list order."""
    sizes = [int(p.numel()) for p in params]
    cumsizes, acc = [], 0
    for s in sizes:
        acc += s
        cumsizes.append(acc)
    return sizes, cumsizes


def _flatten_params(params_with_grad):
    """tf.nest.flatten order (psgd.py:668-669): depth-first through lists/tuples/dicts (dict keys sorted)."""
    if isinstance(params_with_grad, torch.Tensor):
        return [params_with_grad]
    out = []
    if isinstance(params_with_grad, dict):
        for k in sorted(params_with_grad):
            out.extend(_flatten_params(params_with_grad[k]))
    else:
        for p in params_with_grad:
            out.extend(_flatten_params(p))
    return out


class UVd:
    """Low-rank modification (UVd) preconditioner as an optimizer, psgd.py:630-764.

    Same constructor arguments and defaults as the reference (psgd.py:663-666).  Parameters are
    torch tensors with requires_grad=True on a ROCm device (the reference's `.trainable` filter,
    :670, maps to requires_grad).  `step(closure)` returns whatever closure returns (:764)."""

    def __init__(self, params_with_grad, rank_of_modification: int = 10, preconditioner_init_scale=1.0,
                 lr_params=0.01, lr_preconditioner=0.01,
                 grad_clip_max_norm=None, preconditioner_update_probability=1.0,
                 exact_hessian_vector_product: bool = True, generator=None, state_dtype=None, group=None,
                 stage_backend=None, placement="auto"):
        # group (extension): a torch.distributed process group (dist.group.WORLD for the default one) makes this a
        # ROW-SHARDED optimizer: `params_with_grad` are THIS rank's parameters, the global flat vector of psgd.py:729-730 is the
        # concatenation of the ranks' vectors in rank order, and U, V, d hold this rank's rows only.  A step then costs three
        # collectives: the two exchanges of sharded.update_precond_UVd_math_and_precond_grad (r-dimensional sums, never N-sized
        # data) and one scalar all-reduce for the clip norm of :753 (none without clipping); the coins of :703, :562, :588 come
        # from one generator whose state rank 0 broadcasts once.  The closure returns this rank's loss; its gradient and
        # Hessian-vector product with respect to this rank's parameters are what a model-parallel closure computes.
        params = _flatten_params(params_with_grad)
        self._params_with_grad = [p for p in params if p.requires_grad]                      # :670
        p0 = self._params_with_grad[0]
        self._dtype = p0.dtype                                                               # :671
        if self._dtype not in (torch.float32, torch.float16, torch.bfloat16):
            raise TypeError("UVd: parameters must be float32, float16 or bfloat16, got %s" % self._dtype)
        # psgd.py:657-658 allows half-precision parameters.  The preconditioner state U, V, d and all of its arithmetic
        # stay fp32 here (mixed precision: the HIP kernels compute in fp32; v, Hv and the gradient are widened on the
        # way in, the preconditioned gradient is narrowed on the way out).  _tiny and the finite-difference scale follow
        # the parameter dtype as in the reference (:682-683).
        # state_dtype (extension; default None = fp32): "param" or a torch dtype STORES U, V, d in that type, as the reference does
        # for half-precision parameters (psgd.py:688-690) -- every step widens them, runs the fp32 kernels and rounds the updated
        # state back, so the memory held between steps and the rounding of the state once per step are the reference's; its
        # arithmetic (every TF op in the parameters' type) is not reproduced.
        self._state_dtype = torch.float32
        if state_dtype is None:
            self._store_dtype = torch.float32
        else:
            self._store_dtype = self._dtype if state_dtype == "param" else state_dtype
            if self._store_dtype not in (torch.float32, torch.float16, torch.bfloat16):
                raise TypeError("UVd: state_dtype must be None, 'param', float32, float16 or bfloat16, got %r" % (state_dtype,))
        self._device = p0.device
        r = int(rank_of_modification)
        if r < 1:
            raise ValueError("UVd: rank_of_modification must be >= 1, got %d" % r)
        self.lr_params = _Hyper(lr_params)                                                   # :673
        self.lr_preconditioner = _Hyper(lr_preconditioner)                                   # :674
        self.grad_clip_max_norm = _Hyper(math.inf if grad_clip_max_norm is None else grad_clip_max_norm)  # :675-678
        self.preconditioner_update_probability = _Hyper(preconditioner_update_probability)  # :679
        self.exact_hessian_vector_product = _Hyper(bool(exact_hessian_vector_product))       # :680
        self._tiny = torch.finfo(self._dtype).tiny                                           # :682
        self._delta_param_scale = torch.finfo(self._dtype).eps ** 0.5                        # :683
        self._param_sizes, self._param_cumsizes = uvd_param_index(self._params_with_grad)    # :684-685
        num_params = self._param_cumsizes[-1]                                                # :686
        self._generator = generator
        self._group, self._stage_backend, self._num_params_global = group, stage_backend, num_params
        if group is not None:
            from . import sharded as _sharded
            self._sharded = _sharded
            self._num_params_global = _sharded.global_rows(num_params, self._device, group)  # :686 over all ranks (set-up time)
            self._coins = _sharded.branch_rng_for(generator, group, self._device)
        uv_scale = (1.0 / (self._num_params_global * r)) ** 0.5                              # :687 (the GLOBAL N)
        sd = self._state_dtype
        # placement (extension): "probe" / "packed" carve U, V, d, the workspace, the output and the flat v / h / g vectors of :729-730,
        # :747 out of allocations owned by this object (placement.UVdArena; "probe" times candidate layouts once and keeps the
        # fastest: where the WRITTEN streams sit relative to the read ones is worth 5 %); None = plain allocations; "auto" (default)
        # = "probe" when a factor is at least 1 GiB (the search costs ~0.6 s and allocates up to ~100 GiB while it runs; smaller
        # states are close to cache-resident and gain little), None below
        self._arena = None
        if placement not in (None, "auto", "probe", "packed"):
            raise ValueError("UVd: placement must be None, 'auto', 'probe' or 'packed', got %r" % (placement,))
        if placement == "auto":
            placement = "probe" if 4 * num_params * r >= (1 << 30) else None
        if placement is not None and self._device.type == "cuda" and self._store_dtype == torch.float32 \
                and r <= _lib.UVD_MAX_RANK and stage_backend is None:
            from . import placement as _placement
            self._arena = (_placement.UVdArena.probe if placement == "probe" else _placement.UVdArena.packed)(
                num_params, r, self._device)
        if self._arena is not None:
            self._arena.install_workspace()
            self._U, self._V, self._d = self._arena.U, self._arena.V, self._arena.d
            self._U.normal_().mul_(uv_scale)                                                 # :688
            self._V.normal_().mul_(uv_scale)                                                 # :689
            self._d.fill_(float(preconditioner_init_scale))                                  # :690
            return
        self._U = torch.randn(num_params, r, dtype=sd, device=self._device) * uv_scale       # :688
        self._V = torch.randn(num_params, r, dtype=sd, device=self._device) * uv_scale       # :689
        self._d = torch.ones(num_params, 1, dtype=sd, device=self._device) * preconditioner_init_scale  # :690
        if self._store_dtype != sd:
            self._U, self._V, self._d = (x.to(self._store_dtype) for x in (self._U, self._V, self._d))

    def _state_fp32(self):
        """the state the kernels work on: the stored tensors themselves, or fp32 copies of a half-precision state"""
        if self._store_dtype == torch.float32:
            return self._U, self._V, self._d
        return self._U.float(), self._V.float(), self._d.float()

    def _state_store(self, U, V, d):
        if self._store_dtype != torch.float32:
            self._U.copy_(U)
            self._V.copy_(V)
            self._d.copy_(d)

    def _flat(self, tensors, name):
        """psgd.py:729-730, :747: the per-parameter tensors as one flat vector in the state's type -- concatenated straight into
        the arena's region for it when the state is placed (no second copy, and the vector sits where the sweeps read it)"""
        parts = [torch.reshape(x, [-1]) for x in tensors]
        if self._arena is not None and all(x.dtype == self._state_dtype for x in parts):
            return torch.cat(parts, 0, out=getattr(self._arena, name).view(-1))
        return torch.cat(parts, 0).to(self._state_dtype)

    def _loss_of(self, closure_returns):
        return closure_returns if isinstance(closure_returns, torch.Tensor) else closure_returns[0]

    def step(self, closure):
        """psgd.py:692-764."""
        params = self._params_with_grad
        if self._group is None:
            update_Q = _draw_branch(float(self.preconditioner_update_probability), self._generator)   # :703
        else:
            update_Q = self._coins.draw(float(self.preconditioner_update_probability))       # the same coin on every rank
        exact = bool(self.exact_hessian_vector_product)
        vs = None
        if update_Q:
            if exact:                                                                         # :706-714
                with torch.enable_grad():
                    closure_returns = closure()
                    loss = self._loss_of(closure_returns)
                    grads = torch.autograd.grad(loss, params, create_graph=True)
                    vs = [_randn_like(p) for p in params]
                    Hvs = torch.autograd.grad(grads, params, vs)
                grads = [g.detach() for g in grads]
            else:                                                                             # :715-727
                with torch.enable_grad():
                    closure_returns = closure()
                    grads = torch.autograd.grad(self._loss_of(closure_returns), params)
                vs = [_randn_like(p) * self._delta_param_scale for p in params]
                with torch.no_grad():
                    for p, v in zip(params, vs):
                        p.add_(v)
                with torch.enable_grad():
                    perturbed_grads = torch.autograd.grad(self._loss_of(closure()), params)
                Hvs = [pg - g for pg, g in zip(perturbed_grads, grads)]
            v = self._flat(vs, "v")                                                           # :729
            h = self._flat(Hvs, "h")                                                          # :730
            if not exact:                                                                     # :734-736
                v = v / self._delta_param_scale
                h = h / self._delta_param_scale
            grad = self._flat(grads, "g")                                                     # :747
            # :732-733 then :748 as one fused call (same results, three sweeps instead of six)
            U, V, d = self._state_fp32()
            if self._group is None:
                pre_grad = update_precond_UVd_math_and_precond_grad(
                    U, V, d, v[:, None].contiguous(), h[:, None].contiguous(),
                    grad[:, None].contiguous(), step=float(self.lr_preconditioner), tiny=self._tiny,
                    generator=self._generator, out=None if self._arena is None else self._arena.out)
            else:                                  # this rank's rows; 2 exchanges; :562, :588 from the synchronised generator
                pre_grad = self._sharded.update_precond_UVd_math_and_precond_grad(
                    U, V, d, v[:, None].contiguous(), h[:, None].contiguous(),
                    grad[:, None].contiguous(), float(self.lr_preconditioner), self._tiny,
                    generator=self._generator, group=self._group, backend=self._stage_backend,
                    out=None if self._arena is None else self._arena.out)
            self._state_store(U, V, d)
        else:                                                                                 # :737-744
            with torch.enable_grad():
                closure_returns = closure()
                grads = torch.autograd.grad(self._loss_of(closure_returns), params)
            grad = self._flat(grads, "g")                                                     # :747
            if self._group is None:
                pre_grad = precond_grad_UVd_math(*self._state_fp32(), grad[:, None].contiguous(),      # :748
                                                 out=None if self._arena is None else self._arena.out)
            else:
                pre_grad = self._sharded.precond_grad_UVd_math(*self._state_fp32(), grad[:, None].contiguous(),
                                                               group=self._group, backend=self._stage_backend)
        max_norm = float(self.grad_clip_max_norm)
        if math.isinf(max_norm):                                                              # :750-751
            lr = float(self.lr_params)
        else:                                                                                 # :753-754
            if self._group is None:
                grad_norm = torch.sqrt(torch.sum(pre_grad * pre_grad)) + self._tiny
            else:                                  # the norm of the GLOBAL vector: one scalar all-reduce, no host read
                grad_norm = self._sharded.global_norm(pre_grad, self._group) + self._tiny
            lr = float(self.lr_params) * torch.clamp(max_norm / grad_norm, max=1.0)
        with torch.no_grad():                                                                 # :757-762
            undo = (not exact) and update_Q
            for k, (p, i, j) in enumerate(zip(params, self._param_sizes, self._param_cumsizes)):
                delta = (lr * torch.reshape(pre_grad[j - i:j], p.shape)).to(p.dtype)
                if undo:
                    delta = delta + vs[k]
                p.sub_(delta)
        return closure_returns                                                                # :764
