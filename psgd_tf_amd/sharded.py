"""Row-sharded UVd preconditioner across the GPUs of one node (one process per GPU).

The flat parameter vector is split into contiguous row blocks: rank k owns rows
[k N / W, (k+1) N / W) of U, V, d, g, v, h and of the output.  Every sweep is local; the only
exchange is of the tiny reduced buffers between sweeps -- never of N-sized data.  An exchange
is ONE collective: an all-gather (RCCL over xGMI; `backend="nccl"` is RCCL on ROCm) of every
rank's fp64 send region, followed by a local fold of the W copies in rank order (a small HIP
kernel, psgd_*_fold_gathered_f64).  xGMI is point to point, the payloads are 8 B - 30 KB, so
the exchanges are latency-bound: what matters is how many there are per call, and that every
rank ends up with bit-identical reduced values (it folds the same copies in the same order,
whatever algorithm RCCL picks), so the r x r solves and step sizes redone on every rank agree.

    apply  (psgd.py:619-627):  sweep1 -> X(r sums) -> sweep2 -> X(r sums) -> sweep3            2 exchanges
    update (psgd.py:554-617):  [max -> X(2 maxima) -> scale]                                   (balance, :562-567)
                               sweep1 -> X(Gram, <= 30 KB) -> sweep2 -> X(1 maximum) -> sweep3  2 exchanges
    fused update -> apply:     sweep1 -> X(Gram) -> sweep2 -> X([pU | pV | qU | qV | max]) ->
                               r x r algebra -> last sweep (d update + whole apply)             2 exchanges

The two random branches of the reference (psgd.py:562, :588) must agree across ranks as well:
pass them explicitly, or let them be drawn from a branch generator whose state is synchronised
from rank 0 ONCE (first use); afterwards every rank draws the same numbers locally -- no
per-step broadcast and no host synchronisation.

The stage backend is an object with the methods of `HipStages` (the product backend: the C ABI
stage functions of include/psgd_hip.h).  tests/ injects a CPU backend to exercise this
choreography under gloo; there is no CPU backend in the product.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from . import preconditioned_stochastic_gradient_descent as _psgd
from . import uvd_wide as _wide
from . import splu_wide as _splu_wide


class HipStages:
    """Stage functions of the C ABI on this rank's shard; the reduced buffers are views into the
    device workspace (psgd_uvd_ws_region)."""

    def __init__(self, device, n_local, r):
        self.device, self.N, self.r = device, int(n_local), int(r)
        if self.r > _lib.UVD_MAX_RANK:
            raise _lib.PsgdHipError("the stage kernels take ranks up to %d; wider preconditioners go through uvd_wide.py "
                                    "(the sharded entry points route them there), got r = %d" % (_lib.UVD_MAX_RANK, self.r))
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

    def send(self, stage):
        """This rank's fp64 contribution to the exchange after `stage` (a view into the workspace)."""
        return self._view(_lib.PSGD_WS_SEND_F64, stage, torch.float64, 8)

    def gather_buf(self, stage, world):
        key = ("gather", stage, world)
        if key not in self._views:
            self._views[key] = torch.empty(world * self.send(stage).numel(), dtype=torch.float64, device=self.device)
        return self._views[key]

    def fold(self, stage, gathered, world):
        """Fold the all-gathered send regions ([world][count], rank order) into this rank's workspace."""
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_fold_gathered_f64(stage, gathered.data_ptr(), world, self.N, self.r, wp, wn, st),
                   "fold_gathered")

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

    def fused_post(self, step, tiny, update_U):
        """r x r algebra of the fused step on the exchanged sums: s1', s2' of the apply on the updated state."""
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_fused_post_f32(self.N, self.r, float(step), float(tiny), int(bool(update_U)),
                                                    wp, wn, st), "fused_post")

    def fused_final(self, U, V, d, g, step, tiny, out=None):
        """last sweep of the fused step: d update + the whole apply; returns this rank's rows of the result (written to `out` when
        given: placement.UVdArena.out)."""
        out = torch.empty_like(g) if out is None else out
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_fused_final_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), g.data_ptr(),
                                                     out.data_ptr(), self.N, self.r, float(step), float(tiny),
                                                     wp, wn, st), "fused_final")
        return out

    def update_sweep3(self, d, step, tiny):
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_uvd_update_sweep3_f32(d.data_ptr(), self.N, self.r, float(step), float(tiny),
                                                       wp, wn, st), "update_sweep3")


_backends = {}


def hip_backend_for(U):
    if not U.is_cuda:
        raise _lib.PsgdHipError("sharded UVd runs on HIP devices only (tensor is on %s); no CPU fallback" % U.device)
    key = (U.device.index, U.shape[0], U.shape[1], torch.cuda.current_stream(U.device).cuda_stream)
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


def _host_collectives(group):
    """gloo groups reduce host tensors (set-up-time scalars only; the per-step exchanges take device tensors on either backend)"""
    return dist.get_backend(group) == "gloo"


def global_rows(n_local, device, group=None):
    """Rows of the global flat vector = the sum of the ranks' row counts (psgd.py:686 across the group).  Set-up time only:
    one all-reduce and a host read."""
    on_host = _host_collectives(group) or device is None or torch.device(device).type == "cpu"
    t = torch.tensor([int(n_local)], dtype=torch.int64, device="cpu" if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def global_norm(x, group=None):
    """2-norm of the global vector whose rows on this rank are x (the clip norm of psgd.py:753): the local sum of squares in fp64,
    ONE scalar all-reduce (every rank receives the same bits), the root as a device tensor of x's dtype -- no host read."""
    sq = torch.sum(x.to(torch.float64) ** 2).reshape(1)
    comm = _direct_comm(group, sq.device) if sq.is_cuda else None
    if comm is not None:
        comm.all_reduce_sum_f64(sq, torch.cuda.current_stream(sq.device).cuda_stream)
    else:
        dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=group)
    return torch.sqrt(sq[0]).to(x.dtype)


class BranchRng:
    """The two coin flips of update_precond_UVd_math_ (psgd.py:562 p = 0.01, :588 p = 0.5), agreed across ranks
    without a per-step exchange: the state of rank 0's generator is broadcast once, when the object is built (the only
    host synchronisation, at set-up time); afterwards every rank draws the same sequence locally on the host."""

    def __init__(self, generator=None, group=None, device=None):
        src_gen = generator if generator is not None else _psgd._branch_rng
        state = src_gen.get_state().clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            on_host = dist.get_backend(group) == "gloo" or device is None
            buf = state if on_host else state.to(device)
            dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            state = buf.cpu()
        self.gen = torch.Generator(device="cpu")
        self.gen.set_state(state)

    def draw(self, p):
        return bool(torch.rand((), generator=self.gen).item() < p)


_branch_rngs = {}


def branch_rng_for(generator, group, device):
    """One BranchRng per (generator object, group): synchronised from rank 0 on first use, local draws afterwards.
    All ranks must make the same sequence of calls with it (they do: the sharded calls are collective)."""
    key = (id(generator) if generator is not None else None, id(group) if group is not None else None)
    ent = _branch_rngs.get(key)
    if ent is None or ent[0] is not generator:
        ent = (generator, BranchRng(generator, group, device))        # the strong reference keeps id() unique
        _branch_rngs[key] = ent
    return ent[1]


CHECK_EXPLICIT_BRANCHES = False     # debugging aid: verify (one all-reduce + a host read per call) that every rank passed the
                                    # same explicit balance / update_U values; drawn values agree by construction


def _agree_on_branches(balance, update_U, generator, device, group):
    if balance is not None and update_U is not None:
        if CHECK_EXPLICIT_BRANCHES:
            _check_explicit_branches_agree(balance, update_U, device, group)
        return bool(balance), bool(update_U)
    rng = branch_rng_for(generator, group, device)
    b = rng.draw(0.01) if balance is None else bool(balance)            # reference order: :562 then :588
    u = rng.draw(0.5) if update_U is None else bool(update_U)
    return b, u


class _RcclDirect:
    """An RCCL communicator of our own for the exchanges, used through ctypes so that the all-gather is enqueued on the CALLER'S
    stream -- sweep, collective and fold are then one stream-ordered sequence.  torch.distributed runs NCCL / RCCL collectives on a
    stream of the process group: every exchange is a hop there and back (two cross-stream event waits), which measured 26-64 us per
    step on a 1.37-ms step.  The unique id travels over the existing torch.distributed group once, at set-up."""

    class _Uid(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_byte * 128)]

    NCCL_FLOAT64 = 8        # ncclDataType_t

    def __init__(self, group, device):
        import os
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        lib = self.lib
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(self._Uid)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, self._Uid, ctypes.c_int]
        lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        uid = self._Uid()
        if self.rank == 0:
            self._check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        buf = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(device)
        dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ctypes.memmove(ctypes.byref(uid), bytes(buf.cpu().numpy().tobytes()), 128)
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(device):
            self._check(lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")

    def _check(self, rc, what):
        if rc != 0:
            raise _lib.PsgdHipError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def all_gather_f64(self, send, recv, stream):
        """recv[rank k] = rank k's send (fp64, contiguous device tensors), enqueued on `stream` (a raw hipStream_t)"""
        self._check(self.lib.ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), self.NCCL_FLOAT64, self.comm, stream),
                    "ncclAllGather")


    def all_reduce_sum_f64(self, t, stream):
        """in place SUM over the ranks of a contiguous fp64 device tensor, enqueued on `stream`"""
        self._check(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), self.NCCL_FLOAT64, 0, self.comm, stream),
                    "ncclAllReduce")


_direct_comms = {}
EXCHANGES = {"count": 0}    # exchange points passed since import (tests count them per step whatever carries the collective)
DIRECT_RCCL = True      # exchanges of the HIP backend on RCCL groups go through _RcclDirect (PSGD_DIRECT_RCCL=0 or False: torch.distributed)


def _direct_comm(group, device):
    """the _RcclDirect of (group, device), or None when the exchange should go through torch.distributed (gloo groups, the switch
    off, or a set-up failure -- reported once)"""
    import os
    if not DIRECT_RCCL or os.environ.get("PSGD_DIRECT_RCCL", "1") == "0" or dist.get_backend(group) != "nccl":
        return None
    key = (id(group) if group is not None else None, torch.device(device).index, dist.get_world_size(group), dist.get_rank(group))
    if key not in _direct_comms:
        comm, why = None, None
        try:
            comm = _RcclDirect(group, device)
        except Exception as exc:                                            # noqa: BLE001 -- any failure means "use torch.distributed"
            why = exc
        # Every rank must take the SAME route from here on (a rank on torch.distributed and a rank on its own communicator would wait for
        # each other forever): one MIN over "my set-up worked" through the existing group; a single failure anywhere switches all ranks.
        ok = torch.tensor([1.0 if comm is not None else 0.0], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if float(ok.item()) < 1.0:
            if comm is not None:
                why = "another rank's set-up failed"
                try:
                    comm.lib.ncclCommDestroy(comm.comm)
                except Exception:                                            # noqa: BLE001
                    pass
                comm = None
            import warnings
            warnings.warn("psgd_tf_amd.sharded: direct RCCL communicator not available (%s); exchanges use torch.distributed" % (why,))
        _direct_comms[key] = (group, comm)                                  # (the strong reference keeps id() unique)
    return _direct_comms[key][1]


def _exchange(be, stage, group):
    """One exchange point = one collective: all-gather the send regions, fold them in rank order on every rank."""
    EXCHANGES["count"] += 1
    world = dist.get_world_size(group)
    send = be.send(stage)
    gathered = be.gather_buf(stage, world)
    comm = _direct_comm(group, send.device) if send.is_cuda else None
    if comm is not None:
        comm.all_gather_f64(send, gathered, torch.cuda.current_stream(send.device).cuda_stream)
    else:
        dist.all_gather_into_tensor(gathered, send, group=group)
    be.fold(stage, gathered, world)


def _wide_reduce(group):
    """Exchange of the wide-rank path (r > 32, uvd_wide.py): an all-reduce of one small stacked tensor per exchange point.
    Every rank receives the same bits (the collective computes each element once and distributes it).  "summax" (the fused step's
    second exchange: 4r sums and one maximum in one buffer): an all-gather and a fold in rank order on every rank."""
    def reduce(t, op):
        if op == "summax":
            world = dist.get_world_size(group)
            t = t.contiguous()
            gathered = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
            EXCHANGES["count"] += 1
            comm = _direct_comm(group, t.device) if (t.is_cuda and t.dtype == torch.float64) else None
            if comm is not None:
                comm.all_gather_f64(t, gathered, torch.cuda.current_stream(t.device).cuda_stream)
            else:
                dist.all_gather_into_tensor(gathered, t, group=group)
            g2 = gathered.view(world, -1)
            out = g2[0].clone()
            for k in range(1, world):                          # rank order: the same bits on every rank
                out[:-1] += g2[k][:-1]
                out[-1] = torch.maximum(out[-1], g2[k][-1])
            return out
        EXCHANGES["count"] += 1
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=group)
        return t
    return reduce


def _is_wide(U, backend):
    return backend is None and U.dim() == 2 and U.shape[1] > _lib.UVD_MAX_RANK


def _check_local(name, U, V, *cols):
    """The single-GPU entry points' checks on this rank's tensors (device, dtype, contiguity, shapes): the stage kernels
    and the wide-rank chunks take raw pointers.  Only for the product backend (the CPU test backend takes CPU tensors)."""
    _psgd._require_hip(name, U, V, *cols)
    _psgd._uvd_shapes(name, U, V, *cols)


def _check_explicit_branches_agree(balance, update_U, device, group):
    """Explicit branch values must be the same on every rank (drawn ones are, by construction): the ranks would otherwise
    run different stage sequences and hang or mix branches.  One tiny MAX/MIN pair rides on a single all-reduce."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return
    b, u = float(bool(balance)), float(bool(update_U))
    on_host = dist.get_backend(group) == "gloo"
    t = torch.tensor([b, -b, u, -u], dtype=torch.float32, device="cpu" if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    t = t.cpu()
    if float(t[0]) != -float(t[1]) or float(t[2]) != -float(t[3]):
        raise ValueError("sharded UVd: ranks passed different explicit balance / update_U values")


def precond_grad_UVd_math(U, V, d, g, group=None, backend=None):
    """Sharded psgd.py:619-627 on this rank's rows; returns this rank's rows of the result.  2 exchanges."""
    if backend is None:
        _check_local("sharded precond_grad_UVd_math", U, V, d, g)
    if _is_wide(U, backend):
        return _wide.precond_grad(U, V, d, g, _psgd.uvd_workspace, reduce=_wide_reduce(group))
    be = backend if backend is not None else hip_backend_for(U)
    be.apply_sweep1(V, d, g)
    _exchange(be, 1, group)
    be.apply_sweep2(U, d, g)
    _exchange(be, 2, group)
    return be.apply_sweep3(U, V, d, g)


def update_precond_UVd_math_(U, V, d, v, h, step, tiny, *, balance=None, update_U=None, generator=None,
                             group=None, backend=None):
    """Sharded psgd.py:554-617 on this rank's rows (in place, returns None).  2 exchanges (+1 on the balance branch)."""
    if backend is None:
        _check_local("sharded update_precond_UVd_math_", U, V, d, v, h)
    if _is_wide(U, backend):
        balance, update_U = _agree_on_branches(balance, update_U, generator, U.device, group)
        return _wide.update(U, V, d, v, h, float(step), float(tiny), balance, update_U, _psgd.uvd_workspace,
                            reduce=_wide_reduce(group))
    be = backend if backend is not None else hip_backend_for(U)
    balance, update_U = _agree_on_branches(balance, update_U, generator, U.device, group)
    if balance:
        be.balance_max(U, V)
        _exchange(be, 10, group)
        be.balance_scale(U, V)
    be.update_sweep1(U, V, d, v, h)
    _exchange(be, 11, group)
    be.update_sweep2(U, V, d, v, h, step, tiny, update_U)
    _exchange(be, 12, group)
    be.update_sweep3(d, step, tiny)
    return None


def update_precond_UVd_math_and_precond_grad(U, V, d, v, h, g, step, tiny, *, balance=None, update_U=None,
                                             generator=None, group=None, backend=None, out=None):
    """Sharded fused update -> apply; returns this rank's rows of the preconditioned gradient.
    2 exchanges: the Gram; the 4r column sums of sweep 2 with max|nablaD| in one buffer (every rank, r > 32 included since
    round 6)."""
    if backend is None:
        _check_local("sharded update_precond_UVd_math_and_precond_grad", U, V, d, v, h, g)
    if _is_wide(U, backend):
        # r > 32: the fused sequence on the wide-rank building blocks, 2 exchanges (+ 1 on the balance branch) like the ranks below
        balance, update_U = _agree_on_branches(balance, update_U, generator, U.device, group)
        return _wide.update_apply(U, V, d, v, h, g, float(step), float(tiny), balance, update_U, _psgd.uvd_workspace,
                                  reduce=_wide_reduce(group))
    be = backend if backend is not None else hip_backend_for(U)
    balance, update_U = _agree_on_branches(balance, update_U, generator, U.device, group)
    if balance:
        be.balance_max(U, V)
        _exchange(be, 10, group)
        be.balance_scale(U, V)
    be.update_sweep1(U, V, d, v, h)
    _exchange(be, 11, group)
    be.update_sweep2_fused(U, V, d, v, h, g, step, tiny, update_U)
    _exchange(be, 13, group)
    be.fused_post(step, tiny, update_U)
    if out is not None and backend is None:                # (the product backend; a placed state passes its arena's output region)
        return be.fused_final(U, V, d, g, step, tiny, out=out)
    return be.fused_final(U, V, d, g, step, tiny)


# --------------------------------------------------------------------------- sparse LU (psgd.py:396-524), tail rows sharded
class HipSpluStages:
    """Stage functions of the C ABI on this rank's shard of the sparse-LU preconditioner: local tensors
    L12 = [L1; local rows of L2] ([r + n_local, r]), U12 = [U1, local columns of U2], local slices of l3 / u3, flat
    vectors [r corner entries (replicated); local slice of the tail].  Reduced buffers are views into the workspace."""

    def __init__(self, device, n_local_total, r):
        self.device, self.N, self.r = device, int(n_local_total), int(r)
        self.lib = _lib.load()
        self.ws = _psgd._splu_workspace(device, self.N, self.r)
        self._views = {}

    def _w(self):
        return self.ws.data_ptr(), self.ws.numel(), torch.cuda.current_stream(self.device).cuda_stream

    def _view(self, which, stage, dtype, itemsize):
        key = (which, stage)
        if key not in self._views:
            off, cnt = _lib.splu_ws_region(which, stage, self.N, self.r)
            self._views[key] = self.ws[off:off + cnt * itemsize].view(dtype)
        return self._views[key]

    def sums(self, stage):
        return self._view(0, stage, torch.float64, 8)

    def maxbuf(self):
        return self._view(1, 3, torch.float32, 4)

    def send(self, stage):
        return self._view(_lib.PSGD_WS_SEND_F64, stage, torch.float64, 8)

    def gather_buf(self, stage, world):
        key = ("gather", stage, world)
        if key not in self._views:
            self._views[key] = torch.empty(world * self.send(stage).numel(), dtype=torch.float64, device=self.device)
        return self._views[key]

    def fold(self, stage, gathered, world):
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_splu_fold_gathered_f64(stage, gathered.data_ptr(), world, self.N, self.r, wp, wn, st),
                   "splu fold_gathered")

    def stage1(self, U12, x):
        _lib.check(self.lib.psgd_splu_stage1_f32(U12.data_ptr(), x.data_ptr(), self.N, self.r, *self._w()), "splu stage1")

    def apply_stage2(self, L12, l3, U12, u3, g):
        self._out = torch.empty_like(g)
        _lib.check(self.lib.psgd_splu_apply_stage2_f32(L12.data_ptr(), l3.data_ptr(), U12.data_ptr(), u3.data_ptr(),
                                                       g.data_ptr(), self._out.data_ptr(), self.N, self.r, *self._w()),
                   "splu apply stage2")

    def apply_stage3(self, L12, l3, U12, u3):
        out, self._out = self._out, None
        _lib.check(self.lib.psgd_splu_apply_stage3_f32(L12.data_ptr(), l3.data_ptr(), U12.data_ptr(), u3.data_ptr(),
                                                       out.data_ptr(), self.N, self.r, *self._w()), "splu apply stage3")
        return out

    def update_stage2(self, L12, l3, U12, u3, dx, dg):
        _lib.check(self.lib.psgd_splu_update_stage2_f32(L12.data_ptr(), l3.data_ptr(), U12.data_ptr(), u3.data_ptr(),
                                                        dx.data_ptr(), dg.data_ptr(), self.N, self.r, *self._w()),
                   "splu update stage2")

    def update_stage3(self, L12, l3, U12, u3, dx, dg):
        _lib.check(self.lib.psgd_splu_update_stage3_f32(L12.data_ptr(), l3.data_ptr(), U12.data_ptr(), u3.data_ptr(),
                                                        dx.data_ptr(), dg.data_ptr(), self.N, self.r, *self._w()),
                   "splu update stage3")

    def update_stage4(self, L12, l3, U12, u3, dx, dg, step, tiny, has_tail):
        new = [torch.empty_like(t) for t in (L12, l3, U12, u3)]
        wp, wn, st = self._w()
        _lib.check(self.lib.psgd_splu_update_stage4_f32(L12.data_ptr(), l3.data_ptr(), U12.data_ptr(), u3.data_ptr(),
                                                        dx.data_ptr(), dg.data_ptr(), new[0].data_ptr(), new[1].data_ptr(),
                                                        new[2].data_ptr(), new[3].data_ptr(), self.N, self.r, float(step),
                                                        float(tiny), int(bool(has_tail)), wp, wn, st), "splu update stage4")
        return tuple(new)


_splu_backends = {}


def _splu_backend_for(L12):
    if not L12.is_cuda:
        raise _lib.PsgdHipError("sharded sparse LU runs on HIP devices only (tensor is on %s); no CPU fallback" % L12.device)
    key = (L12.device.index, L12.shape[0], L12.shape[1], torch.cuda.current_stream(L12.device).cuda_stream)
    if key not in _splu_backends:
        _splu_backends[key] = HipSpluStages(L12.device, L12.shape[0], L12.shape[1])
    return _splu_backends[key]


def _splu_is_wide(L12, backend):
    return backend is None and L12.dim() == 2 and L12.shape[1] > _lib.SPLU_MAX_RANK


def precond_grad_splu(L12, l3, U12, u3, grad, group=None, backend=None):
    """Sharded psgd.py:483-524.  `grad` is this rank's flat vector [r corner entries; local tail slice]; returns the
    same layout (corner entries identical on every rank).  2 exchanges (r sums; 2r sums of which r are used).  Ranks above 32
    (the reference has no limit, psgd.py:420) run on column chunks (splu_wide.py) with an all-reduce per exchange."""
    if _splu_is_wide(L12, backend):
        _psgd._require_hip("sharded precond_grad_splu", L12, l3, U12, u3, grad)
        return _splu_wide.precond_grad(L12, l3, U12, u3, grad.reshape(-1), _psgd.uvd_workspace,
                                       reduce=_wide_reduce(group)).reshape(grad.shape)
    be = backend if backend is not None else _splu_backend_for(L12)
    be.stage1(U12, grad)
    _exchange(be, 1, group)
    be.apply_stage2(L12, l3, U12, u3, grad)
    _exchange(be, 2, group)
    return be.apply_stage3(L12, l3, U12, u3)


def update_precond_splu(L12, l3, U12, u3, dx, dg, step=0.01, tiny=None, has_tail=True, group=None, backend=None):
    """Sharded psgd.py:396-480 on this rank's rows; returns this rank's (L12, l3, U12, u3) (corner blocks identical on
    every rank).  3 exchanges: r sums, 2r sums, [r sums | 4 maxima].  has_tail: the global problem has tail rows."""
    tiny = _psgd._tiny if tiny is None else tiny
    if _splu_is_wide(L12, backend):
        _psgd._require_hip("sharded update_precond_splu", L12, l3, U12, u3, dx, dg)
        return _splu_wide.update(L12, l3, U12, u3, dx.reshape(-1), dg.reshape(-1), float(step), float(tiny), _psgd.uvd_workspace,
                                 reduce=_wide_reduce(group))
    be = backend if backend is not None else _splu_backend_for(L12)
    be.stage1(U12, dg)
    _exchange(be, 1, group)
    be.update_stage2(L12, l3, U12, u3, dx, dg)
    _exchange(be, 2, group)
    be.update_stage3(L12, l3, U12, u3, dx, dg)
    _exchange(be, 3, group)
    return be.update_stage4(L12, l3, U12, u3, dx, dg, step, tiny, has_tail)
