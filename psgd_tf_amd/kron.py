"""Kronecker-product preconditioner: shape dispatch (psgd.py:72-152) and the seven formats.

(dense, dense) -- the hot path, psgd.py:156-192 -- runs in the HIP kernels of
csrc/psgd_kron.hip through the C ABI.  The sparse formats (normalization / scaling
factors, psgd.py:198-391) are reachable through the same two public entry
points and run in HIP as well: elementwise / reduction kernels for the sparse half, the same
MFMA GEMM and triangular solve for the dense half of a mixed format.
"""
import collections
import weakref

import torch

from . import _lib

_tiny = torch.finfo(torch.float32).tiny
_f32 = torch.float32
try:                                        # the current stream's handle without building a torch.cuda.Stream object
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                      # pragma: no cover
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream

KRON_FORMATS = ("dense_dense", "dense_norm", "dense_scale", "norm_dense",
                "norm_scale", "scale_dense", "scale_norm", "unknown")


def kron_format(shape_l, shape_r):
    """The dispatch table of psgd.py:80-110 / :122-152.  Square is tested first, so a
    [1,1] or [2,2] factor is dense (README.md:39)."""
    m, n = int(shape_l[0]), int(shape_l[1])
    p, q = int(shape_r[0]), int(shape_r[1])
    if m == n:
        return "dense_dense" if p == q else "dense_norm" if p == 2 else "dense_scale" if p == 1 else "unknown"
    if m == 2:
        return "norm_dense" if p == q else "norm_scale" if p == 1 else "unknown"
    if m == 1:
        return "scale_dense" if p == q else "scale_norm" if p == 2 else "unknown"
    return "unknown"


def _check_rank2_f32(name, *tensors, allow_bf16_last=False, allow_bf16_from=None):
    # the reference pins its public Kron functions to rank-2 fp32 (psgd.py:67-71, 113-115);
    # the one extension is bf16 data for (dense, dense): a bf16 gradient for the apply (BASELINE config 5),
    # a bf16 (dX, dG) pair for the update.  The factors stay fp32.
    first_bf16 = len(tensors) - 1 if allow_bf16_last else allow_bf16_from
    f32 = torch.float32
    for i, t in enumerate(tensors):
        if t.dim() != 2:
            raise ValueError("%s: rank-2 tensors required, got shape %s" % (name, tuple(t.shape)))
        if t.dtype is not f32 and not (first_bf16 is not None and i >= first_bf16 and t.dtype == torch.bfloat16):
            raise TypeError("%s: fp32 tensors required, got %s" % (name, t.dtype))


def _check_kron_shapes(name, Ql, Qr, *mats):
    """Every format keeps M = rows(data) in the last dimension of the left factor ((M,M), (2,M) or (1,M)) and
    N = cols(data) in the last dimension of the right one.  The reference's matmuls raise on a mismatch; the kernels
    take raw pointers and would read or write out of bounds, so the check is here."""
    shape = mats[0].shape
    M, N = shape
    for t in mats[1:]:
        if t.shape != shape:
            raise ValueError("%s: data matrices must share one shape, got %s and %s"
                             % (name, tuple(mats[0].shape), tuple(t.shape)))
    if Ql.shape[1] != M or Qr.shape[1] != N:
        raise ValueError("%s: factors %s, %s do not match the %d x %d data matrix (left factor needs %d columns, "
                         "right factor %d)" % (name, tuple(Ql.shape), tuple(Qr.shape), M, N, M, N))
    dev = mats[0].device
    for t in (Ql, Qr) + tuple(mats[1:]):
        if t.device != dev:
            raise ValueError("%s: all tensors must be on one device, got %s and %s" % (name, dev, t.device))


# Prepared factor-only state (the Grams of the fp32 apply, the bf16 factor copies, padded factors) is reused by the
# next apply when the factor tensors are the very objects it was made from, at the same version and address.  That is
# exact for the reference's call pattern (update_precond_kron returns NEW tensors, mnist_with_lenet5.py:51-52) and for
# version-tracked in-place ops (Q.mul_(), Q.copy_(), Q[...] = ...).  It cannot see writes that bypass the version
# counter: `Q.data.mul_()`, DLPack / raw-pointer kernels writing into the same storage.  Callers that do that must call
# invalidate_factor_cache() after the write, or switch the reuse off with set_factor_cache(False).  The reuse is also
# off while the current stream is being captured into a graph: the prepared-or-not decision would be frozen into the
# graph, and replays after an in-graph factor update would read stale Grams.
_factor_cache_enabled = True
_factor_cache_epoch = 0


def set_factor_cache(enabled):
    """Switch the reuse of prepared factor-only state across applies on or off (default on).  Returns the old value."""
    global _factor_cache_enabled
    old, _factor_cache_enabled = _factor_cache_enabled, bool(enabled)
    invalidate_factor_cache()
    return old


def invalidate_factor_cache():
    """Forget every prepared Gram / bf16 factor copy / padded factor: the next apply of every shape rebuilds them.  Call
    after modifying factor tensors in a way the version counter does not see (`Q.data`, external kernels)."""
    global _factor_cache_epoch
    _factor_cache_epoch += 1
    _prepared.clear()
    _bf16_prepared.clear()
    _padded_factors.clear()
    for slot in _apply_slots.values():
        slot.rl = slot.rr = None
        slot.prepared = False


def set_tuning(key, value):
    """psgd_kron_set_tuning through the Python boundary: keys that change what the prepared state in a workspace means
    (1: plane products or the in-GEMM split, 4: operand planes or not, 12: their format, 16: exact or bound-based scales -- the
    C side gates the prepared form on all of them: fp32 Grams against planes) also drop every prepared Gram / plane set, as the
    header requires."""
    _lib.check(_lib.load().psgd_kron_set_tuning(int(key), int(value)), "psgd_kron_set_tuning")
    if int(key) in (1, 4, 12, 16):
        invalidate_factor_cache()
        _apply_slots.clear()                         # (which apply paths a shape has is asked when its slot is made)


# --------------------------------------------------------------------------- independent per-layer calls on forked streams
_layer_ctx = None
_layer_pool = {}          # device index -> list of streams (made once, reused by every block)


class layer_streams:
    """`with kron.layer_streams():` around the reference's per-layer list comprehensions (mnist_with_lenet5.py:51, :53).

    The update_precond_kron / precond_grad_kron calls inside the block must be INDEPENDENT of each other (one per layer: no
    call reads what another call of the same block returns).  Each runs on the next stream of a small pool that first waits
    for the caller's stream; leaving the block makes the caller's stream wait for all of them, so everything after the
    block sees the results in stream order.  Small layers are bound by their chain of 3 (apply) / 5 (update) dependent
    launches, not by work, so the chains of different layers overlap.  Event fork/join only: inside `torch.cuda.graph` the
    captured graph gets one branch per call and a replay runs the layers side by side with no host cost at all.
    Workspaces are per (shape, stream), so two layers of one shape never share scratch.  One block at a time, from one host
    thread (the switch is a module global, like the factor cache)."""

    def __init__(self, streams=8):
        self.n = int(streams)
        if self.n < 1:
            raise ValueError("layer_streams: streams must be >= 1")

    def __enter__(self):
        global _layer_ctx
        if _layer_ctx is not None or _batch_ctx is not None:
            raise RuntimeError("layer_streams / layer_batch blocks do not nest")
        self.main = None
        self.used = []
        self.keep = []
        self.count = 0
        _layer_ctx = self
        return self

    def __exit__(self, *exc):
        global _layer_ctx
        _layer_ctx = None
        for s in self.used:
            self.main.wait_stream(s)
        self.keep = []
        return False

    def run(self, fn, args):
        global _layer_ctx
        first = args[0]
        if not (torch.is_tensor(first) and first.is_cuda):       # (not a device call: let the entry point raise its own error, unforked)
            _layer_ctx = None
            try:
                return fn(*args)
            finally:
                _layer_ctx = self
        if self.main is None:
            self.main = torch.cuda.current_stream(first.device)
            pool = _layer_pool.setdefault(first.get_device(), [])
            while len(pool) < self.n:
                pool.append(torch.cuda.Stream(device=first.device))
            self.pool = pool
        s = self.pool[self.count % self.n]
        self.count += 1
        # EVERY call waits for what the caller's stream holds at this point, not only a stream's first call of the block: with more
        # calls than streams (any model with more than `streams` layers) a later call's arguments may have been produced on the
        # caller's stream after the pool stream last waited (`precond_grad_kron(ql, qr, g * scale)` inside the comprehension).
        s.wait_stream(self.main)
        if s not in self.used:
            self.used.append(s)
        # ... and its arguments stay alive until the block joins: a temporary allocated on the caller's stream and dropped when the
        # call returns must not be handed out again (to the next allocation on that stream) while the side stream still reads it
        self.keep.append(args)
        for t in args:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(s)
        _layer_ctx = None                            # (the call itself must not fork again)
        try:
            with torch.cuda.stream(s):
                out = fn(*args)
        finally:
            _layer_ctx = self
        for t in (out if isinstance(out, (tuple, list)) else (out,)):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.main)           # allocated on the side stream, consumed on the caller's
        return out


# --------------------------------------------------------------------------- route of the large fp32 apply
_apply_route = "reference"


def set_apply_route(route):
    """Which chain the fp32 (dense, dense) apply of a LARGE layer runs.  Returns the old value.

    "reference" (default since round 6): always the association order of psgd.py:189-192 -- the Gram of the smaller side, then
        the products of :190 / :192 -- whatever was seen before: identical inputs give identical bits on every call (the prepared
        Gram is reused when the factors are unchanged; it is the same Gram, bit for bit, that a fresh call would make).
    "auto" (opt-in fast path): factors seen for the first time take the Gram-free chain Ql'(Ql((G Qr')Qr)), the same factor
        tensors a second time make the Gram of psgd.py:190 / :192 and later calls reuse it -- fastest for the reference's
        update -> apply pattern (4096^2: 0.9 ms against 1.03), but calls 1, 2 and 3 on identical inputs return bits that differ
        in the last places (3e-6 between the routes, tests/test_kron_gpu.py), and for M >= N the first-sight association order
        is not the reference's."""
    global _apply_route
    if route not in ("auto", "reference"):
        raise ValueError("set_apply_route: 'auto' or 'reference', got %r" % (route,))
    old, _apply_route = _apply_route, route
    return old


# --------------------------------------------------------------------------- deferred batching of independent per-layer calls
_batch_ctx = None
_BATCH_MAX_DIM = 512          # layers the batched kernels are for (launch-bound); larger ones fill the chip and run at once


class layer_batch:
    """`with kron.layer_batch():` around the reference's per-layer list comprehensions (mnist_with_lenet5.py:51, :53).

    Inside the block update_precond_kron / precond_grad_kron on small fp32 (dense, dense) layers (M, N <= 512) do not launch
    anything: they allocate and return their output tensors and queue a descriptor; leaving the block issues ONE batched call
    per kind (psgd_kron_dd_{update,apply}_batched_f32: 5 / 3 launches for the whole set instead of per layer) that fills them.
    Same one-line intrusion and the same requirement as layer_streams: the calls of one comprehension are independent of each
    other, and nothing reads a returned tensor before the block ends.  A call of the other kind flushes what is queued first, so
    `Qs = [update ...]` followed by `[apply with the new Qs ...]` in ONE block is in order.  Results are bit for bit those of the
    batched entry points (= the per-layer calls).  Other formats, bf16 operands and larger layers run at once, as outside the
    block.  Capturable: the block then records the batched launches."""

    def __enter__(self):
        global _batch_ctx
        if _batch_ctx is not None or _layer_ctx is not None:
            raise RuntimeError("layer_batch / layer_streams blocks do not nest")
        self.kind, self.items, self.step = None, [], None
        _batch_ctx = self
        return self

    def __exit__(self, exc_type, *exc):
        global _batch_ctx
        _batch_ctx = None
        if exc_type is None:
            self.flush()
        else:
            # the block is left by an exception: the queued calls are dropped, and the tensors they already returned must not be
            # readable as results -- NaN, not whatever the allocator left there
            for it in self.items:
                for o in (it[-1] if isinstance(it[-1], tuple) else (it[-1],)):
                    o.fill_(float("nan"))
            self.items, self.kind = [], None
        return False

    def flush(self):
        global _batch_ctx
        items, kind, self.items, self.kind = self.items, self.kind, [], None
        if not items:
            return
        saved, _batch_ctx = _batch_ctx, None
        try:
            if kind == "update":
                if len(items) == 1:
                    a, b, x, g, o = items[0]
                    _dd_update_f32(a, b, x, g, self.step, x.shape[0], x.shape[1], outs=o)
                else:
                    update_precond_kron_batched([i[0] for i in items], [i[1] for i in items], [i[2] for i in items],
                                                [i[3] for i in items], self.step, outs=[i[4] for i in items])
            else:
                if len(items) == 1:
                    a, b, g, o = items[0]
                    _dd_apply_f32(a, b, g, g.shape[0], g.shape[1], out=o)
                else:
                    precond_grad_kron_batched([i[0] for i in items], [i[1] for i in items], [i[2] for i in items],
                                              outs=[i[3] for i in items])
        finally:
            _batch_ctx = saved

    def update(self, Ql, Qr, dX, dG, step):
        if self.kind not in (None, "update") or (self.items and step != self.step):
            self.flush()
        self.kind, self.step = "update", step
        outs = (torch.empty((Ql.shape[0], Ql.shape[1]), dtype=_f32, device=Ql.device),
                torch.empty((Qr.shape[0], Qr.shape[1]), dtype=_f32, device=Qr.device))
        self.items.append((Ql, Qr, dX, dG, outs))
        return outs

    def apply(self, Ql, Qr, Grad):
        if self.kind not in (None, "apply"):
            self.flush()
        self.kind = "apply"
        out = torch.empty((Grad.shape[0], Grad.shape[1]), dtype=_f32, device=Grad.device)
        self.items.append((Ql, Qr, Grad, out))
        return out


def _cache_usable():
    return _factor_cache_enabled and not torch.cuda.is_current_stream_capturing()


def _version_of(t):
    try:
        return t._version
    except RuntimeError:              # inference-mode tensors do not track versions: never a hit
        return None


class _FactorTag:
    """Identity of the factor tensors some prepared, factor-only data (Grams, bf16 copies) in a workspace was made
    from: the very tensor objects (weak references), their version counters and their storage pointers.  A new tensor
    that the allocator placed at the address of a freed one is a different object, hence a miss."""

    def __init__(self, tensors):
        self.refs = [weakref.ref(t) for t in tensors]
        self.meta = [(_version_of(t), t.data_ptr()) for t in tensors]

    def matches(self, tensors):
        return len(tensors) == len(self.refs) and all(r() is t and m[0] is not None and m == (_version_of(t), t.data_ptr())
                                                      for r, m, t in zip(self.refs, self.meta, tensors))


_prepared = {}            # workspace key -> _FactorTag of the Grams it holds (fp32 apply, single and batched)


def _is_prepared(key, tensors):
    tag = _prepared.get(key)
    return tag is not None and tag.matches(tensors) and _cache_usable()


# --------------------------------------------------------------------------- dense (x) dense: HIP
_kron_ws = _lib.WorkspaceCache()
_kron_ws_bf16 = _lib.WorkspaceCache()


def _stream_key(device):
    """Workspaces carry intermediates between the launches of one call: one per (shape, stream)."""
    return torch.cuda.current_stream(device).cuda_stream


def _kron_workspace(device, M, N):
    key = (device.index if device.index is not None else torch.cuda.current_device(), M, N, _stream_key(device))
    def make():
        nbytes = int(_lib.load().psgd_kron_dd_workspace_bytes(M, N))
        if nbytes < 0:
            _lib.check(nbytes, "psgd_kron_dd_workspace_bytes")
        return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
    return _kron_ws.get(key, make)


def _require_hip(name, *tensors):
    for t in tensors:
        if not t.is_cuda:
            raise _lib.PsgdHipError("%s runs on the HIP device only (tensor is on %s); no CPU fallback" % (name, t.device))


def _pad8(n):
    return (n + 7) // 8 * 8


def _bf16_pad_shape(M, N, apply):
    """Padded extents of a bf16-operand problem.  Multiples of 8 always (16-byte chunks along every K axis).  The apply
    goes up to multiples of 256 when that makes at least 16 tiles of 256 x 256 and adds at most a fifth to the elements:
    only then do the 256 x 256 kernels (fused triangular pairs, 8-phase dense kernel) take it, and they are up to 1.5x
    faster than the guarded 128-tile kernels on the unpadded shape (2500^2: 0.30 -> 0.17 ms; tools/kron_shape_scan.py)."""
    Mp, Np = _pad8(M), _pad8(N)
    if apply:
        M2, N2 = (M + 255) // 256 * 256, (N + 255) // 256 * 256
        if (M2 // 256) * (N2 // 256) >= 16 and M2 * N2 <= 1.2 * M * N:      # (the 256-tile pairs take >= 16 tiles)
            Mp, Np = M2, N2
    return Mp, Np


# (device, M, N, Mp, Np, stream) -> (tag of the original factors, padded Ql, padded Qr); least-recently-used, a few entries
# (each holds two padded fp32 factors: a program that sweeps shapes must not accumulate them)
_padded_factors = collections.OrderedDict()
_PADDED_FACTORS_MAX = 8


def _padded_bf16_problem(Ql, Qr, mats, apply=False):
    """The bf16 kernels move 16-byte chunks (8 elements) along every K axis.  Other shapes run zero-padded
    (_bf16_pad_shape): factors become blockdiag(Q, c I) (c = tiny for the apply, max|diag Q| for the update: triangular,
    invertible at the factor's own scale, and it cannot win the max of psgd.py:166-167), data matrices get zero rows
    and columns.  Every product and solve of psgd.py:156-192 is then block diagonal: the leading M x N (M x M, N x N)
    block of each result is the unpadded result, the rest is zero (or tiny I).  For the apply the padded factors are
    kept while the original ones are unchanged (same tensor objects and versions), so that their bf16 copies in the
    workspace stay valid from call to call as they do for unpadded shapes."""
    M, N = mats[0].shape
    Mp, Np = _bf16_pad_shape(M, N, apply)

    def factor(Q, n, n_p):
        if n == n_p:
            return Q
        Qp = torch.zeros(n_p, n_p, dtype=Q.dtype, device=Q.device)
        Qp[:n, :n] = Q
        # apply: the pad block only ever multiplies zero data, any finite value does.  update: the pad diagonal goes
        # through the balance (x rho or / rho, psgd.py:166-170) and is then INVERTED by the solves of :174, so it must
        # be of the factor's own magnitude -- tiny / rho overflowed to inf for rho >= 4 and turned both new factors
        # into NaN.  max|diag Q| cannot raise the max of :166-167 (it is one of the entries the max runs over) and the
        # pad block stays decoupled: its rows of A and Bt are zero, so its gradient block is zero.
        Qp.diagonal()[n:] = _tiny if apply else Q.diagonal().abs().max()
        return Qp
    pads = [torch.nn.functional.pad(x, (0, Np - N, 0, Mp - M)) for x in mats]
    if not apply:
        return factor(Ql, M, Mp), factor(Qr, N, Np), pads
    key = (Ql.device.index, M, N, Mp, Np, _stream_key(Ql.device))
    hit = _padded_factors.get(key)
    if hit is not None and hit[0].matches((Ql, Qr)) and _cache_usable():
        _padded_factors.move_to_end(key)
        return hit[1], hit[2], pads
    Qlp, Qrp = factor(Ql, M, Mp), factor(Qr, N, Np)
    _padded_factors[key] = (_FactorTag((Ql, Qr)), Qlp, Qrp)
    _padded_factors.move_to_end(key)
    while len(_padded_factors) > _PADDED_FACTORS_MAX:
        _padded_factors.popitem(last=False)
    return Qlp, Qrp, pads


def _update_precond_dense_dense_bf16(Ql, Qr, dX, dG, step):
    """psgd.py:156-179 with bf16 MFMA operands for the products (dX, dG in bf16; fp32 master factors in and out;
    balance, triangular solves, norms and the final subtraction stay fp32)."""
    M, N = dX.shape
    if M % 8 or N % 8:
        Qlp, Qrp, (dXp, dGp) = _padded_bf16_problem(Ql, Qr, (dX, dG))
        a, b = _update_precond_dense_dense_bf16(Qlp, Qrp, dXp, dGp, step)
        return a[:M, :M].contiguous(), b[:N, :N].contiguous()
    Ql, Qr, dX, dG = (t.contiguous() for t in (Ql, Qr, dX, dG))
    QlO, QrO = torch.empty_like(Ql), torch.empty_like(Qr)
    key = (dX.device.index, "upd", M, N, _stream_key(dX.device))
    ws = _kron_ws_bf16.get(key, lambda: torch.empty(int(_lib.load().psgd_kron_dd_update_workspace_bytes_bf16(M, N)),
                                                      dtype=torch.uint8, device=dX.device))
    rc = _lib.load().psgd_kron_dd_update_bf16(Ql.data_ptr(), Qr.data_ptr(), dX.data_ptr(), dG.data_ptr(),
                                               QlO.data_ptr(), QrO.data_ptr(), M, N, float(step), float(_tiny),
                                               ws.data_ptr(), ws.numel(),
                                               torch.cuda.current_stream(dX.device).cuda_stream)
    _lib.check(rc, "psgd_kron_dd_update_bf16")
    return QlO, QrO


def _update_precond_dense_dense(Ql, Qr, dX, dG, step):
    """psgd.py:156-179 on the GPU; pure (fresh outputs)."""
    _require_hip("update_precond_kron", Ql, Qr, dX, dG)
    if dX.dtype != dG.dtype:
        raise TypeError("update_precond_kron: dX and dG must share a dtype, got %s and %s" % (dX.dtype, dG.dtype))
    if dX.dtype == torch.bfloat16:
        return _update_precond_dense_dense_bf16(Ql, Qr, dX, dG, step)
    return _dd_update_f32(Ql, Qr, dX, dG, step, dX.shape[0], dX.shape[1])


def _dd_update_f32(Ql, Qr, dX, dG, step, M, N, outs=None):
    """fp32 dense (x) dense update on checked device tensors (see _dd_apply_f32); shares the slot (workspace) of the apply.
    outs: (QlOut, QrOut) to fill instead of fresh tensors (layer_batch)."""
    if not Ql.is_contiguous():
        Ql = Ql.contiguous()
    if not Qr.is_contiguous():
        Qr = Qr.contiguous()
    if not dX.is_contiguous():
        dX = dX.contiguous()
    if not dG.is_contiguous():
        dG = dG.contiguous()
    idx = dX.get_device()
    st = _raw_stream(idx)
    key = (idx, M, N, st)
    slot = _apply_slots.get(key)
    if slot is None or slot.ws() is None:
        slot = _apply_slots[key] = _ApplySlot(_kron_workspace(dX.device, M, N), M, N)
    else:
        _kron_ws.touch(key)                          # (same key as _kron_workspace: the hot shape must not age out of the LRU)
    QlO, QrO = outs if outs is not None else (torch.empty_like(Ql), torch.empty_like(Qr))
    rc = slot.fn_update(Ql.data_ptr(), Qr.data_ptr(), dX.data_ptr(), dG.data_ptr(), QlO.data_ptr(), QrO.data_ptr(), M, N,
                        float(step), _tiny, slot.ws_ptr, slot.ws_bytes, st)
    if rc:
        _lib.check(rc, "psgd_kron_dd_update_f32")
    return QlO, QrO


_bf16_apply_shapes = {}
_bf16_prepared = {}       # workspace key -> (weakref Ql, weakref Qr, versions, data pointers) of the factor copies it holds


def check_bf16_handoffs():
    """Diagnostics of the fused triangular pairs of the bf16 precond_grad_kron: how many tile hand-offs so far ran into
    their wait bound (~0.5 s: possible only when other streams or processes hold CUs that long while the call runs).
    Results are unaffected -- the waiting workgroup produces the missing tile itself and redoes its accumulation, the
    call returns the undisturbed bits -- but every such recovery costs up to the bound, so a non-zero count on a
    persistently shared device is a reason to select the kernels without in-launch hand-offs
    (psgd_kron_bf16_set_tuning(0, 4)).  Synchronises with the device: call it where the host waits anyway.  Returns the
    total count over all bf16 apply workspaces."""
    total = 0
    for key, (M, N) in list(_bf16_apply_shapes.items()):
        if key in _kron_ws_bf16:
            ws = _kron_ws_bf16[key]
            rc = _lib.load().psgd_kron_bf16_handoff_timeouts(ws.data_ptr(), M, N)
            if rc < 0:
                _lib.check(rc, "psgd_kron_bf16_handoff_timeouts")
            total += rc
    return total


# ---- automatic switch to the hand-off-free kernels (round 6).  The fused pair recovers from a hand-off whose producer was not
# resident in time, but every recovery costs up to its wait bound (~0.5 s): on a device that is persistently shared each call can
# pay it.  Every HANDOFF_CHECK_EVERY-th bf16 apply on a workspace copies the workspace's recovery counter to pinned host memory
# WITHOUT synchronising (the copy rides on the stream; its result is read a later call, once its event has completed); after
# HANDOFF_FALLBACK_AFTER recoveries on one workspace the library switches -- for the process -- to the kernels without in-launch
# hand-offs (psgd_kron_bf16_set_tuning(0, 4)) and says so once.  bf16_handoff_fallback_active() tells; reset_bf16_handoff_fallback()
# goes back (e.g. when the other tenant has left).
HANDOFF_CHECK_EVERY = 16
HANDOFF_FALLBACK_AFTER = 3
_handoff_watch = {}       # workspace key -> [calls, counter view, pinned host word, event or None, count at reset]
_handoff_fallback = [False]


def bf16_handoff_fallback_active():
    return _handoff_fallback[0]


def reset_bf16_handoff_fallback():
    """Back to the fused pairs (the default route) after an automatic fall-back; the recovery counts start from zero."""
    if _handoff_fallback[0]:
        _lib.check(_lib.load().psgd_kron_bf16_set_tuning(0, 0), "psgd_kron_bf16_set_tuning")
        _handoff_fallback[0] = False
    _handoff_watch.clear()


def _watch_handoffs(key, ws, M, N, device):
    w = _handoff_watch.get(key)
    if w is None or w[1].data_ptr() < ws.data_ptr() or w[1].data_ptr() >= ws.data_ptr() + ws.numel():
        off = int(_lib.load().psgd_kron_bf16_handoff_counter_offset(M, N))
        if off < 0:
            return
        w = _handoff_watch[key] = [0, ws[off:off + 4].view(torch.int32), torch.zeros(1, dtype=torch.int32).pin_memory(), None, None]
    w[0] += 1
    if w[3] is not None and w[3].query():                  # an earlier copy has landed: look at it
        seen = int(w[2][0])
        w[3] = None
        if w[4] is None:
            w[4] = seen                                     # (counts from before the watch started do not count)
        elif seen - w[4] >= HANDOFF_FALLBACK_AFTER and not _handoff_fallback[0]:
            import warnings
            _lib.check(_lib.load().psgd_kron_bf16_set_tuning(0, 4), "psgd_kron_bf16_set_tuning")
            _handoff_fallback[0] = True
            warnings.warn("psgd_tf_amd.kron: %d hand-offs of the fused bf16 pairs ran into their wait bound on this device (other "
                          "work holds its CUs): switched to the kernels without in-launch hand-offs "
                          "(kron.reset_bf16_handoff_fallback() goes back)" % (seen - w[4]))
    if w[3] is None and (w[0] % HANDOFF_CHECK_EVERY == 1) and not torch.cuda.is_current_stream_capturing():
        w[2].copy_(w[1], non_blocking=True)
        w[3] = torch.cuda.Event()
        w[3].record(torch.cuda.current_stream(device))


def _precond_grad_dense_dense_bf16(Ql, Qr, Grad):
    """psgd.py:182-192 with bf16 MFMA operands (Grad and result in bf16, fp32 master factors)."""
    M, N = Grad.shape
    if _bf16_pad_shape(M, N, True) != (M, N):
        Qlp, Qrp, (Gp,) = _padded_bf16_problem(Ql, Qr, (Grad,), apply=True)
        return _precond_grad_dense_dense_bf16(Qlp, Qrp, Gp)[:M, :N].contiguous()
    Ql, Qr, Grad = (t.contiguous() for t in (Ql, Qr, Grad))      # (.contiguous() returns the same object when it already is)
    out = torch.empty_like(Grad)
    key = (Grad.device.index, M, N, _stream_key(Grad.device))

    def make():
        _bf16_prepared.pop(key, None)
        ws = torch.empty(int(_lib.load().psgd_kron_dd_workspace_bytes_bf16(M, N)), dtype=torch.uint8, device=Grad.device)
        _lib.check(_lib.load().psgd_kron_bf16_handoff_reset(ws.data_ptr(), M, N, _stream_key(Grad.device)),
                   "psgd_kron_bf16_handoff_reset")
        _bf16_apply_shapes[key] = (M, N)
        return ws
    ws = _kron_ws_bf16.get(key, make)
    lib, st = _lib.load(), torch.cuda.current_stream(Grad.device).cuda_stream
    # The bf16 copies of the factors live in the workspace and only change when the factors do: convert again only if
    # these are not the very tensor objects (same storage, same version counter) the copies were made from.
    tag = _bf16_prepared.get(key)
    if not (tag is not None and tag.matches((Ql, Qr)) and _cache_usable()):
        # new factors: ONE call converts them and applies (the conversion launch zeroes the pairs' hand-off flags: no memset launch)
        rc = lib.psgd_kron_dd_apply_bf16(Ql.data_ptr(), Qr.data_ptr(), Grad.data_ptr(), out.data_ptr(), M, N, ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "psgd_kron_dd_apply_bf16")
        _bf16_prepared[key] = _FactorTag((Ql, Qr))
    else:
        rc = lib.psgd_kron_dd_apply_bf16_prepared(Grad.data_ptr(), out.data_ptr(), M, N, ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "psgd_kron_dd_apply_bf16_prepared")
    if not _handoff_fallback[0]:
        _watch_handoffs(key, ws, M, N, Grad.device)
    return out


class _ApplySlot:
    """Per (device, shape, stream): the workspace and what its prepared Grams were made from.  Small layers are
    host-bound (a launch costs the host ~4 us, the GPU less), so the per-call Python work is kept to a handful of
    attribute reads: the factor identity check is spelled out instead of built from generators."""
    __slots__ = ("ws", "ws_ptr", "ws_bytes", "rl", "rr", "vl", "vr", "pl", "pr", "fn_apply", "fn_prepared", "fn_update", "fn_direct",
                 "prepared", "path")

    def __init__(self, ws, M, N):
        # (a weak reference: the workspace cache owns the block; a slot of a shape the cache has evicted must not keep it alive)
        self.ws, self.ws_ptr, self.ws_bytes = weakref.ref(ws), ws.data_ptr(), ws.numel()
        self.rl = self.rr = None
        self.vl = self.vr = self.pl = self.pr = -1
        lib = _lib.load()
        self.fn_apply, self.fn_prepared = lib.psgd_kron_dd_apply_f32, lib.psgd_kron_dd_apply_prepared_f32
        self.fn_update = lib.psgd_kron_dd_update_f32
        # large layers: factors seen for the first time go through the Gram-free chain (nothing prepared); the Gram is made when
        # the SAME factors come a second time.  fn_direct is None where that chain is not a path of its own (small layers).
        self.fn_direct = lib.psgd_kron_dd_apply_direct_f32 if lib.psgd_kron_dd_apply_direct_distinct(M, N) == 1 else None
        self.prepared = False
        self.path = ""                               # which of "direct" / "both" / "prepared" the last apply took (tests)


_apply_slots = {}


def _precond_grad_dense_dense(Ql, Qr, Grad):
    """psgd.py:182-192 on the GPU."""
    _require_hip("precond_grad_kron", Ql, Qr, Grad)
    if Grad.dtype == torch.bfloat16:
        return _precond_grad_dense_dense_bf16(Ql, Qr, Grad)
    return _dd_apply_f32(Ql, Qr, Grad, Grad.shape[0], Grad.shape[1])


def _dd_apply_f32(Ql, Qr, Grad, M, N, out=None):
    """fp32 dense (x) dense apply on device tensors whose shapes, dtypes and devices the caller has checked.  Small layers
    are host-bound (LeNet5: a launch costs the GPU ~4 us, the reference's per-layer call pattern five calls per step), so
    this path is a handful of attribute reads, one allocation and ONE ctypes call."""
    if not Ql.is_contiguous():
        Ql = Ql.contiguous()
    if not Qr.is_contiguous():
        Qr = Qr.contiguous()
    if not Grad.is_contiguous():
        Grad = Grad.contiguous()
    idx = Grad.get_device()
    st = _raw_stream(idx)
    key = (idx, M, N, st)
    slot = _apply_slots.get(key)
    if slot is None or slot.ws() is None:            # first call for this (device, shape, stream), or the cache evicted its block
        slot = _apply_slots[key] = _ApplySlot(_kron_workspace(Grad.device, M, N), M, N)
        if len(_apply_slots) > 4 * _kron_ws.max_entries:     # slots of evicted workspaces: drop them
            for k in [k for k, v in _apply_slots.items() if v.ws() is None]:
                del _apply_slots[k]
    else:
        _kron_ws.touch(key)
    if out is None:
        out = torch.empty_like(Grad)
    pl, pr = Ql.data_ptr(), Qr.data_ptr()
    # factor-only half (the Grams, kept in the workspace): redone only when these are not the very factor tensors (same
    # objects, same version counters, same storage) it was made from
    vl, vr = _version_of(Ql), _version_of(Qr)
    if (slot.rl is not None and slot.rl() is Ql and slot.rr() is Qr and slot.vl == vl and slot.vr == vr and vl is not None
            and vr is not None and slot.pl == pl and slot.pr == pr and _cache_usable()):
        if slot.prepared:
            rc = slot.fn_prepared(pl, pr, Grad.data_ptr(), out.data_ptr(), M, N, slot.ws_ptr, slot.ws_bytes, st)
            if rc:
                _lib.check(rc, "psgd_kron_dd_apply_prepared_f32")
            slot.path = "prepared"
        else:                                        # the same factors a second time: now their Gram pays (prepare + apply_prepared)
            rc = slot.fn_apply(pl, pr, Grad.data_ptr(), out.data_ptr(), M, N, slot.ws_ptr, slot.ws_bytes, st)
            if rc:
                _lib.check(rc, "psgd_kron_dd_apply_f32")
            slot.prepared = True
            slot.path = "both"
    else:
        if slot.fn_direct is not None and _apply_route == "auto":      # new factors, large layer: the Gram-free chain, nothing prepared
            rc = slot.fn_direct(pl, pr, Grad.data_ptr(), out.data_ptr(), M, N, slot.ws_ptr, slot.ws_bytes, st)
            if rc:
                _lib.check(rc, "psgd_kron_dd_apply_direct_f32")
            slot.prepared = False
            slot.path = "direct"
        else:                                        # both halves in one call (= prepare + apply_prepared)
            rc = slot.fn_apply(pl, pr, Grad.data_ptr(), out.data_ptr(), M, N, slot.ws_ptr, slot.ws_bytes, st)
            if rc:
                _lib.check(rc, "psgd_kron_dd_apply_f32")
            slot.prepared = True
            slot.path = "both"
        slot.rl, slot.rr = weakref.ref(Ql), weakref.ref(Qr)
        slot.vl, slot.vr, slot.pl, slot.pr = vl, vr, pl, pr
    return out


# --------------------------------------------------------------------------- batched (dense, dense): HIP
_batch_ws = _lib.WorkspaceCache()


def _batched_ok(Qls, Qrs, mats):
    return all(kron_format(a.shape, b.shape) == "dense_dense" and a.dtype == b.dtype == g.dtype == torch.float32
               and g.is_cuda for a, b, g in zip(Qls, Qrs, mats))


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(vals):
    import ctypes
    return (ctypes.c_int * len(vals))(*vals)


def _batch_workspace(device, Ms, Ns):
    key = (device.index, tuple(Ms), tuple(Ns), _stream_key(device))
    def make():
        _prepared.pop(("ddb",) + key, None)
        nbytes = int(_lib.load().psgd_kron_dd_workspace_bytes_batched(_int_array(Ms), _int_array(Ns), len(Ms)))
        if nbytes < 0:
            _lib.check(nbytes, "psgd_kron_dd_workspace_bytes_batched")
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    return _batch_ws.get(key, make)


def precond_grad_kron_batched(Qls, Qrs, Grads, outs=None):
    """[precond_grad_kron(Ql, Qr, G) for ...] (mnist_with_lenet5.py:53) with the same stage of every
    (dense, dense) layer in one kernel launch.  Other formats fall back to the per-layer call.
    outs: contiguous fp32 tensors to fill (layer_batch; only with lists the batched kernels take)."""
    Qls, Qrs, Grads = list(Qls), list(Qrs), list(Grads)
    if not (len(Qls) == len(Qrs) == len(Grads)):
        raise ValueError("precond_grad_kron_batched: the three lists must have one length")
    if not _batched_ok(Qls, Qrs, Grads):
        res = [precond_grad_kron(a, b, g) for a, b, g in zip(Qls, Qrs, Grads)]
        if outs is not None:                         # (a caller that handed out `outs` already must find the results there)
            for o, r_ in zip(outs, res):
                o.copy_(r_)
            return list(outs)
        return res
    for a, b, g in zip(Qls, Qrs, Grads):
        _check_rank2_f32("precond_grad_kron_batched", a, b, g)
        _check_kron_shapes("precond_grad_kron_batched", a, b, g)
    Qls, Qrs, Grads = ([t.contiguous() for t in ts] for ts in (Qls, Qrs, Grads))
    if outs is None:
        outs = [torch.empty_like(g) for g in Grads]
    Ms, Ns = [g.shape[0] for g in Grads], [g.shape[1] for g in Grads]
    dev = Grads[0].device
    ws = _batch_workspace(dev, Ms, Ns)
    lib, st = _lib.load(), torch.cuda.current_stream(dev).cuda_stream
    key = ("ddb", dev.index, tuple(Ms), tuple(Ns), st)
    pl, pr, pm, pn = _ptr_array(Qls), _ptr_array(Qrs), _int_array(Ms), _int_array(Ns)
    if _is_prepared(key, Qls + Qrs):                 # the Grams of every layer's factors are in the workspace
        rc = lib.psgd_kron_dd_apply_prepared_batched_f32(pl, pr, _ptr_array(Grads), _ptr_array(outs), pm, pn, len(Ms),
                                                         ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "psgd_kron_dd_apply_prepared_batched_f32")
    else:                                            # Grams (one launch) + gradient half, one call
        rc = lib.psgd_kron_dd_apply_batched_f32(pl, pr, _ptr_array(Grads), _ptr_array(outs), pm, pn, len(Ms),
                                                ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "psgd_kron_dd_apply_batched_f32")
        _prepared[key] = _FactorTag(Qls + Qrs)
    return outs


def update_precond_kron_batched(Qls, Qrs, dXs, dGs, step=0.01, outs=None):
    """[update_precond_kron(Ql, Qr, dX, dG, step) for ...] (mnist_with_lenet5.py:51), batched as above.
    Returns a list of (Ql_new, Qr_new).  outs: list of (QlOut, QrOut) to fill (layer_batch; all layers <= 512 only)."""
    Qls, Qrs, dXs, dGs = list(Qls), list(Qrs), list(dXs), list(dGs)
    if not (len(Qls) == len(Qrs) == len(dXs) == len(dGs)):
        raise ValueError("update_precond_kron_batched: the four lists must have one length")
    def _fill(res):                                  # (a caller that handed out `outs` already must find the results there)
        if outs is None:
            return res
        for (ol, orr), (a, b) in zip(outs, res):
            ol.copy_(a)
            orr.copy_(b)
        return [tuple(o) for o in outs]
    if not (_batched_ok(Qls, Qrs, dXs) and _batched_ok(Qls, Qrs, dGs)):
        return _fill([update_precond_kron(a, b, x, g, step) for a, b, x, g in zip(Qls, Qrs, dXs, dGs)])
    small = [i for i, x in enumerate(dXs) if max(x.shape) <= 512]
    if len(small) < len(dXs):
        # a mixed list (the LSTM / NMT drivers mix sizes across 512): the small layers -- launch-bound -- still share their
        # launches, the large ones -- each fills the chip -- go one by one
        out = [None] * len(dXs)
        if len(small) > 1:
            for i, res in zip(small, update_precond_kron_batched([Qls[i] for i in small], [Qrs[i] for i in small],
                                                                  [dXs[i] for i in small], [dGs[i] for i in small], step)):
                out[i] = res
        for i in range(len(dXs)):
            if out[i] is None:
                out[i] = update_precond_kron(Qls[i], Qrs[i], dXs[i], dGs[i], step)
        return _fill(out)
    for a, b, x, g in zip(Qls, Qrs, dXs, dGs):
        _check_rank2_f32("update_precond_kron_batched", a, b, x, g)
        _check_kron_shapes("update_precond_kron_batched", a, b, x, g)
    Qls, Qrs, dXs, dGs = ([t.contiguous() for t in ts] for ts in (Qls, Qrs, dXs, dGs))
    if outs is not None:
        QlO, QrO = [o[0] for o in outs], [o[1] for o in outs]
    else:
        QlO, QrO = [torch.empty_like(t) for t in Qls], [torch.empty_like(t) for t in Qrs]
    Ms, Ns = [x.shape[0] for x in dXs], [x.shape[1] for x in dXs]
    dev = dXs[0].device
    ws = _batch_workspace(dev, Ms, Ns)
    rc = _lib.load().psgd_kron_dd_update_batched_f32(_ptr_array(Qls), _ptr_array(Qrs), _ptr_array(dXs), _ptr_array(dGs),
                                                      _ptr_array(QlO), _ptr_array(QrO), _int_array(Ms), _int_array(Ns),
                                                      len(Ms), float(step), float(_tiny), ws.data_ptr(), ws.numel(),
                                                      torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "psgd_kron_dd_update_batched_f32")
    return list(zip(QlO, QrO))


# --------------------------------------------------------------------------- sparse formats: HIP
# Canonical orientations (psgd.py:198-391).  The data matrices may be transposed views (the mirrored
# formats of the dispatcher pass dX.t(), dG.t(), Grad.t()): their strides go straight to the kernels.
_SPARSE_FMT = {"ds": 0, "nd": 1, "ns": 2}
_sparse_ws = _lib.WorkspaceCache()


def _sparse_workspace(device, fmt, M, N):
    key = (device.index, fmt, M, N, _stream_key(device))
    def make():
        nbytes = int(_lib.load().psgd_kron_sparse_workspace_bytes(_SPARSE_FMT[fmt], M, N))
        if nbytes < 0:
            _lib.check(nbytes, "psgd_kron_sparse_workspace_bytes")
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    return _sparse_ws.get(key, make)


def _sparse_update(fmt, L, R, dX, dG, step):
    _require_hip("update_precond_kron", L, R, dX, dG)
    M, N = dX.shape
    # The kernels take strided views, but a transposed view (the mirrored formats) makes every elementwise pass and column
    # reduction read with a stride of M floats: one tiled transpose up front (torch) is ~30x cheaper than that at
    # embedding sizes.  C-ABI callers may still pass views.
    dX, dG = dX.contiguous(), dG.contiguous()
    L, R = L.contiguous(), R.contiguous()
    Lo, Ro = torch.empty_like(L), torch.empty_like(R)
    ws = _sparse_workspace(dX.device, fmt, M, N)
    fn = getattr(_lib.load(), "psgd_kron_%s_update_f32" % fmt)
    rc = fn(L.data_ptr(), R.data_ptr(), dX.data_ptr(), dG.data_ptr(), dX.stride(0), dX.stride(1), Lo.data_ptr(),
            Ro.data_ptr(), M, N, float(step), float(_tiny), ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream(dX.device).cuda_stream)
    _lib.check(rc, "psgd_kron_%s_update_f32" % fmt)
    return Lo, Ro


def _sparse_apply(fmt, L, R, Grad):
    _require_hip("precond_grad_kron", L, R, Grad)
    M, N = Grad.shape
    Grad = Grad.contiguous()                                     # (see _sparse_update)
    L, R = L.contiguous(), R.contiguous()
    out = torch.empty(M, N, dtype=Grad.dtype, device=Grad.device)
    ws = _sparse_workspace(Grad.device, fmt, M, N)
    fn = getattr(_lib.load(), "psgd_kron_%s_apply_f32" % fmt)
    rc = fn(L.data_ptr(), R.data_ptr(), Grad.data_ptr(), Grad.stride(0), Grad.stride(1), out.data_ptr(), M, N,
            ws.data_ptr(), ws.numel(), torch.cuda.current_stream(Grad.device).cuda_stream)
    _lib.check(rc, "psgd_kron_%s_apply_f32" % fmt)
    return out


def _update_precond_norm_dense(ql, Qr, dX, dG, step):
    """psgd.py:198-246."""
    return _sparse_update("nd", ql, Qr, dX, dG, step)


def _precond_grad_norm_dense(ql, Qr, Grad):
    """psgd.py:249-270."""
    return _sparse_apply("nd", ql, Qr, Grad)


def _update_precond_dense_scale(Ql, qr, dX, dG, step):
    """psgd.py:276-307."""
    return _sparse_update("ds", Ql, qr, dX, dG, step)


def _precond_grad_dense_scale(Ql, qr, Grad):
    """psgd.py:310-322."""
    return _sparse_apply("ds", Ql, qr, Grad)


def _update_precond_norm_scale(ql, qr, dX, dG, step):
    """psgd.py:328-369."""
    return _sparse_update("ns", ql, qr, dX, dG, step)


def _precond_grad_norm_scale(ql, qr, Grad):
    """psgd.py:372-391."""
    return _sparse_apply("ns", ql, qr, Grad)


# --------------------------------------------------------------------------- public dispatchers
def _is_dd_f32(Ql, Qr, X):
    """fp32 dense (x) dense operands of consistent shapes on one ROCm device: exactly what _check_rank2_f32,
    _check_kron_shapes and _require_hip establish for that format, as one expression (the per-layer call pattern of
    mnist_with_lenet5.py:51,53 is host-bound at LeNet5 sizes)."""
    sl, sr, sx = Ql.shape, Qr.shape, X.shape
    return (len(sx) == 2 and len(sl) == 2 and len(sr) == 2 and sl[0] == sl[1] == sx[0] and sr[0] == sr[1] == sx[1]
            and X.dtype is _f32 and Ql.dtype is _f32 and Qr.dtype is _f32 and X.is_cuda
            and Ql.device == X.device and Qr.device == X.device)


# ---- the triangular contract, checked on request.  psgd.py:173, :179, :190, :192 multiply with the FULL Ql, Qr; the kernels assume
# upper-triangular factors (large layers skip whole tiles below the diagonal, small layers and diagonal tiles multiply what is stored).
# All agree for every factor the reference can produce (identity initialisation, updates that preserve triangularity), so the default
# costs nothing; a caller that builds factors itself can switch the check on and gets a ValueError instead of a silently different
# result (one reduction and a host read per dense factor per call).
_check_triangular = [False]


def set_triangular_check(on):
    """True: update_precond_kron / precond_grad_kron verify that square (dense) factors have an all-zero strictly lower triangle."""
    old, _check_triangular[0] = _check_triangular[0], bool(on)
    return old


def _assert_upper(name, *factors):
    for q in factors:
        if q.dim() == 2 and q.shape[0] == q.shape[1] and q.shape[0] > 1 and bool(torch.count_nonzero(torch.tril(q, -1))):
            raise ValueError("%s: a dense factor has non-zero entries below its diagonal; the kernels treat factors as upper "
                             "triangular (the reference's own factors always are: psgd.py:40-42, :175-179)" % name)


def update_precond_kron(Ql, Qr, dX, dG, step=0.01):
    if _check_triangular[0]:
        _assert_upper("update_precond_kron", Ql, Qr)
    if _layer_ctx is not None:
        return _layer_ctx.run(update_precond_kron, (Ql, Qr, dX, dG, step))
    if _is_dd_f32(Ql, Qr, dX) and dG.shape == dX.shape and dG.dtype is _f32 and dG.device == dX.device:
        if _batch_ctx is not None:
            if dX.shape[0] <= _BATCH_MAX_DIM and dX.shape[1] <= _BATCH_MAX_DIM:
                return _batch_ctx.update(Ql, Qr, dX, dG, step)
            _batch_ctx.flush()                                                              # (keeps the issue order of the block)
        return _dd_update_f32(Ql, Qr, dX, dG, step, dX.shape[0], dX.shape[1])                # psgd.py:84
    if _batch_ctx is not None:
        _batch_ctx.flush()
    fmt = kron_format(Ql.shape, Qr.shape)
    _check_rank2_f32("update_precond_kron", Ql, Qr, dX, dG, allow_bf16_from=(2 if fmt == "dense_dense" else None))
    if fmt != "unknown":
        _check_kron_shapes("update_precond_kron", Ql, Qr, dX, dG)
    if fmt == "dense_dense":
        return _update_precond_dense_dense(Ql, Qr, dX, dG, step)                            # psgd.py:84
    if fmt == "dense_norm":
        return _update_precond_norm_dense(Qr, Ql, dX.t(), dG.t(), step)[::-1]               # :86
    if fmt == "dense_scale":
        return _update_precond_dense_scale(Ql, Qr, dX, dG, step)                            # :88
    if fmt == "norm_dense":
        return _update_precond_norm_dense(Ql, Qr, dX, dG, step)                             # :94
    if fmt == "norm_scale":
        return _update_precond_norm_scale(Ql, Qr, dX, dG, step)                             # :96
    if fmt == "scale_dense":
        return _update_precond_dense_scale(Qr, Ql, dX.t(), dG.t(), step)[::-1]              # :102
    if fmt == "scale_norm":
        return _update_precond_norm_scale(Qr, Ql, dX.t(), dG.t(), step)[::-1]               # :104
    print("Unknown Kronecker product preconditioner, no update")                            # :90,98,106,109
    return Ql, Qr


def precond_grad_kron(Ql, Qr, Grad):
    if _check_triangular[0]:
        _assert_upper("precond_grad_kron", Ql, Qr)
    if _layer_ctx is not None:
        return _layer_ctx.run(precond_grad_kron, (Ql, Qr, Grad))
    if _is_dd_f32(Ql, Qr, Grad):
        if _batch_ctx is not None:
            if Grad.shape[0] <= _BATCH_MAX_DIM and Grad.shape[1] <= _BATCH_MAX_DIM:
                return _batch_ctx.apply(Ql, Qr, Grad)
            _batch_ctx.flush()
        return _dd_apply_f32(Ql, Qr, Grad, Grad.shape[0], Grad.shape[1])                     # psgd.py:126
    if _batch_ctx is not None:
        _batch_ctx.flush()
    fmt = kron_format(Ql.shape, Qr.shape)
    _check_rank2_f32("precond_grad_kron", Ql, Qr, Grad, allow_bf16_last=(fmt == "dense_dense"))
    if fmt != "unknown":
        _check_kron_shapes("precond_grad_kron", Ql, Qr, Grad)
    if fmt == "dense_dense":
        return _precond_grad_dense_dense(Ql, Qr, Grad)                                      # psgd.py:126
    if fmt == "dense_norm":
        return _precond_grad_norm_dense(Qr, Ql, Grad.t()).t()                               # :128
    if fmt == "dense_scale":
        return _precond_grad_dense_scale(Ql, Qr, Grad)                                      # :130
    if fmt == "norm_dense":
        return _precond_grad_norm_dense(Ql, Qr, Grad)                                       # :136
    if fmt == "norm_scale":
        return _precond_grad_norm_scale(Ql, Qr, Grad)                                       # :138
    if fmt == "scale_dense":
        return _precond_grad_dense_scale(Qr, Ql, Grad.t()).t()                              # :144
    if fmt == "scale_norm":
        return _precond_grad_norm_scale(Qr, Ql, Grad.t()).t()                               # :146
    print("Unknown Kronecker product preconditioner, no preconditioning")                   # :132,140,148,151
    return Grad
