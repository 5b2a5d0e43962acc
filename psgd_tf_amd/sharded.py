"""Row-sharded UVd preconditioner across the GPUs of one node (one process per GPU).

The flat parameter vector is split into contiguous row blocks: rank k owns rows
[k N / W, (k+1) N / W) of U, V, d, g, v, h and of the output.  Every sweep is local; the only
exchange is an all-reduce (RCCL over xGMI; `backend="nccl"` is RCCL on ROCm) of the tiny
reduced buffers between sweeps -- never of N-sized data:

    apply  (psgd.py:619-627):  sweep1 -> SUM r fp64 -> sweep2 -> SUM r fp64 -> sweep3
    update (psgd.py:554-617):  [max -> MAX 2 fp32 -> scale]                      (balance, :562-567)
                               sweep1 -> SUM Gram fp64 (<= 30 KB) -> sweep2 -> MAX 1 fp32 -> sweep3

All ranks then hold bit-identical reduced values, so the r x r solves and step sizes computed
redundantly on every rank agree.  The two random branches of the reference (psgd.py:562, :588)
must agree as well: pass them explicitly or let rank 0 draw and broadcast them.

The stage backend is an object with the methods of `HipStages` (the product backend: the C ABI
stage functions of include/psgd_hip.h).  tests/ injects a CPU backend to exercise this
choreography under gloo; there is no CPU backend in the product.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from . import preconditioned_stochastic_gradient_descent as _psgd


class HipStages:
    """Stage functions of the C ABI on this rank's shard; the reduced buffers are views into the
    device workspace (psgd_uvd_ws_region)."""

    def __init__(self, device, n_local, r):
        self.device, self.N, self.r = device, int(n_local), int(r)
        self.lib = _lib.load()
        self.ws = _psgd.uvd_workspace(device, self.N, self.r)
        self._views = {}

    def _st(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _view(self, which, stage, dtype, itemsize):
        key = (which, stage)
        if key not in self._views:
            off, cnt = _lib.ws_region(which, stage, self.N, self.r)
            self._views[key] = self.ws[off:off + cnt * itemsize].view(dtype)
        return self._views[key]

    def sums(self, stage):
        return self._view(_lib.PSGD_WS_SUMS_F64, stage, torch.float64, 8)

    def maxbuf(self, stage):
        return self._view(_lib.PSGD_WS_MAX_F32, stage, torch.float32, 4)

    def _w(self):
        return self.ws.data_ptr(), self.ws.numel(), self._st()

    def apply_sweep1(self, V, d, g):
        _lib.check(self.lib.psgd_uvd_apply_sweep1_f32(V.data_ptr(), d.data_ptr(), g.data_ptr(), self.N, self.r,
                                                      *self._w()), "apply_sweep1")

    def apply_sweep2(self, U, d, g):
        self._out = torch.empty_like(g)           # holds g1 = d.*g + U s1 until sweep 3
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_apply_sweep2_f32(U.data_ptr(), d.data_ptr(), g.data_ptr(), self._out.data_ptr(),
                                                      self.N, self.r, 1, wp, wn, st), "apply_sweep2")

    def apply_sweep3(self, U, V, d, g):
        out, self._out = self._out, None
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_apply_sweep3_f32(V.data_ptr(), d.data_ptr(), out.data_ptr(), self.N, self.r, 1,
                                                      wp, wn, st), "apply_sweep3")
        return out

    def balance_max(self, U, V):
        _lib.check(self.lib.psgd_uvd_balance_max_f32(U.data_ptr(), V.data_ptr(), self.N, self.r, *self._w()),
                   "balance_max")

    def balance_scale(self, U, V):
        _lib.check(self.lib.psgd_uvd_balance_scale_f32(U.data_ptr(), V.data_ptr(), self.N, self.r, *self._w()),
                   "balance_scale")

    def update_sweep1(self, U, V, d, v, h):
        _lib.check(self.lib.psgd_uvd_update_sweep1_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(),
                                                       h.data_ptr(), self.N, self.r, *self._w()), "update_sweep1")

    def update_sweep2(self, U, V, d, v, h, step, tiny, update_U):
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_update_sweep2_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(),
                                                       h.data_ptr(), self.N, self.r, float(step), float(tiny),
                                                       int(bool(update_U)), wp, wn, st), "update_sweep2")

    def update_sweep2_fused(self, U, V, d, v, h, g, step, tiny, update_U):
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_update_sweep2_fused_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(),
                                                             h.data_ptr(), g.data_ptr(), self.N, self.r, float(step),
                                                             float(tiny), int(bool(update_U)), wp, wn, st),
                   "update_sweep2_fused")

    def fused_s1(self, step, tiny):
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_fused_s1_f32(self.N, self.r, float(step), float(tiny), wp, wn, st), "fused_s1")

    def apply_sweep2_local_s1(self, U, d, g):
        """apply sweep 2 when s1 is already in place on every rank (after fused_s1)."""
        self._out = torch.empty_like(g)
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_apply_sweep2_f32(U.data_ptr(), d.data_ptr(), g.data_ptr(), self._out.data_ptr(),
                                                      self.N, self.r, 0, wp, wn, st), "apply_sweep2")

    def update_sweep3(self, d, step, tiny):
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_update_sweep3_f32(d.data_ptr(), self.N, self.r, float(step), float(tiny),
                                                       wp, wn, st), "update_sweep3")


_backends = {}


def hip_backend_for(U):
    if not U.is_cuda:
        raise _lib.PsgdHipError("sharded UVd runs on HIP devices only (tensor is on %s); no CPU fallback" % U.device)
    key = (U.device.index, U.shape[0], U.shape[1])
    if key not in _backends:
        _backends[key] = HipStages(U.device, U.shape[0], U.shape[1])
    return _backends[key]


def shard_rows(n_global, rank, world):
    """Contiguous row block of rank `rank`: [lo, hi).  Block starts are multiples of 64 rows: every shard of an
    aligned [N, r] array stays 16-byte aligned for any r, and the per-row vectors (d, g, v, h) of a shard start on a
    256-byte boundary, so the tiles of the sweeps cover whole 128-byte lines."""
    per = -(-n_global // world)
    per = (per + 63) // 64 * 64
    lo = min(rank * per, n_global)
    hi = min(lo + per, n_global)
    return lo, hi


def _agree_on_branches(balance, update_U, generator, device, group):
    if balance is not None and update_U is not None:
        return bool(balance), bool(update_U)
    flags = torch.zeros(2, dtype=torch.float32, device=device)
    if dist.get_rank(group) == 0:
        b = _psgd._draw_branch(0.01, generator) if balance is None else bool(balance)
        u = _psgd._draw_branch(0.5, generator) if update_U is None else bool(update_U)
        flags = torch.tensor([float(b), float(u)], dtype=torch.float32, device=device)
    dist.broadcast(flags, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    b, u = flags.tolist()
    return bool(b), bool(u)


def precond_grad_UVd_math(U, V, d, g, group=None, backend=None):
    """Sharded psgd.py:619-627 on this rank's rows; returns this rank's rows of the result."""
    be = backend if backend is not None else hip_backend_for(U)
    be.apply_sweep1(V, d, g)
    dist.all_reduce(be.sums(1), op=dist.ReduceOp.SUM, group=group)
    be.apply_sweep2(U, d, g)
    dist.all_reduce(be.sums(2), op=dist.ReduceOp.SUM, group=group)
    return be.apply_sweep3(U, V, d, g)


def update_precond_UVd_math_(U, V, d, v, h, step, tiny, *, balance=None, update_U=None, generator=None,
                             group=None, backend=None):
    """Sharded psgd.py:554-617 on this rank's rows (in place, returns None)."""
    be = backend if backend is not None else hip_backend_for(U)
    balance, update_U = _agree_on_branches(balance, update_U, generator, U.device, group)
    if balance:
        be.balance_max(U, V)
        dist.all_reduce(be.maxbuf(10), op=dist.ReduceOp.MAX, group=group)
        be.balance_scale(U, V)
    be.update_sweep1(U, V, d, v, h)
    dist.all_reduce(be.sums(11), op=dist.ReduceOp.SUM, group=group)
    be.update_sweep2(U, V, d, v, h, step, tiny, update_U)
    dist.all_reduce(be.maxbuf(12), op=dist.ReduceOp.MAX, group=group)
    be.update_sweep3(d, step, tiny)
    return None


def update_precond_UVd_math_and_precond_grad(U, V, d, v, h, g, step, tiny, *, balance=None, update_U=None,
                                             generator=None, group=None, backend=None):
    """Sharded fused update -> apply; returns this rank's rows of the preconditioned gradient.
    Exchanges: SUM Gram, MAX max|nablaD| + SUM 2r (p, q), SUM r (s2) -- one all-reduce fewer than update + apply."""
    be = backend if backend is not None else hip_backend_for(U)
    balance, update_U = _agree_on_branches(balance, update_U, generator, U.device, group)
    if balance:
        be.balance_max(U, V)
        dist.all_reduce(be.maxbuf(10), op=dist.ReduceOp.MAX, group=group)
        be.balance_scale(U, V)
    be.update_sweep1(U, V, d, v, h)
    dist.all_reduce(be.sums(11), op=dist.ReduceOp.SUM, group=group)
    be.update_sweep2_fused(U, V, d, v, h, g, step, tiny, update_U)
    dist.all_reduce(be.maxbuf(12), op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(be.sums(13), op=dist.ReduceOp.SUM, group=group)
    be.update_sweep3(d, step, tiny)
    be.fused_s1(step, tiny)
    be.apply_sweep2_local_s1(U, d, g)
    dist.all_reduce(be.sums(2), op=dist.ReduceOp.SUM, group=group)
    return be.apply_sweep3(U, V, d, g)
