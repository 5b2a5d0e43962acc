"""UVd preconditioner of rank r > 32 (the reference has no rank limit, psgd.py:663).

Round 5 -- ranks 33 .. 64 on one GPU, contiguous 16-byte aligned U and V: whole-matrix entry points of the C ABI, no chunks --
    precond_grad          psgd_uvd_wide_apply_cols_f32        (three sweeps per four columns of g)
    update                psgd_uvd_wide_update_f32            (one-sweep Gram, r x r block, sweep 2, d: four launches)
    update + precond_grad psgd_uvd_wide_update_apply_f32      (the UVd.step pattern: U and V read three times)
row-sharded runs of those ranks use the whole-matrix building blocks (psgd_uvd_gram_wide_f32, psgd_uvd_wide_colsums_f32,
psgd_uvd_wide_axpy_cols_f32, psgd_uvd_wide_rank2_update_f32) with one exchange per stage.  Everything else -- ranks above 64,
strided views, unaligned matrices, the A/B switches PSGD_WIDE_FULL / PSGD_WIDE_UPDATE / PSGD_WIDE_STEP = 0 -- takes the route of
rounds 3-4 described below.

The rank-templated sweep kernels of that route exist for r = 1..32.  A wider preconditioner is handled on column chunks:
U = [U_1 | ... | U_c], V = [V_1 | ... | V_c], every chunk an [N, rc] matrix, rc <= 32.  When r splits evenly (r = c rc with
c = ceil(r / 32) or one more: 40, 48, 50, 64, 96, 100, 128 ...) the chunks are column VIEWS of U and V -- row stride r, the
`*_ld` entry points of the C ABI (round 4) -- and nothing is copied; otherwise contiguous copies, the last one padded with zero
columns (which change nothing: U V' is the same product and K = I + V'U only gains identity rows).  Every N-sized pass runs in
the HIP kernels through the C ABI:

    Gram blocks X'Y, X't, X'w      psgd_uvd_update_sweep1_f32 on pairs of chunks (matrix cores), one call per pair
    S = M' [x_0 ..]                psgd_uvd_colsums_f32           (psgd.py:544 inner product)
    out_j = x_j + M S_j            psgd_uvd_axpy_cols_f32         (psgd.py:544 outer product; two vectors per sweep)
    M <- M - (a c1' - b c2')       psgd_uvd_rank2_update_f32      (psgd.py:600-601 / :614-615)

What is left to torch is what has no N x r extent: gathering / scattering the column chunks (slicing), elementwise
operations on N-vectors (d .* h, v ./ d, nablaD, the d update) and the r x r algebra (two fp64 solves with K, the norm
of psgd.py:594-596 / :608-610) -- all on the device, nothing synchronises.  Same formulas, same order of operations on
the state as psgd.py:554-627; costs (2c - 1) passes over U and V for the Gram instead of one, so r <= 32 stays on the
specialised path.

Row-sharded use (psgd_tf_amd/sharded.py, round 3): every function takes `reduce(tensor, "sum" | "max")`, applied to the
quantities that are sums or maxima over the rows -- all column sums of one stage stacked into ONE tensor, all Gram blocks
into one, the maxima of |U|, |V| (balance branch) and of |nablaD| -- so a sharded apply costs 2 exchanges and a sharded update
2 (+ 1 on the balance branch), like the specialised path.  The default reduces nothing (one GPU holds all rows).
"""
import ctypes
import os

import torch

from . import _lib

_MAXR = _lib.UVD_MAX_RANK
_gram_scratch = {}        # (device, N, r) -> scratch of the wide Gram kernel (a few entries)
_update_scratch = {}      # (device, stream, N, r) -> scratch of psgd_uvd_wide_update_f32 (holds nablaD: about 4 N bytes)


def _chunks(r):
    """(chunks, chunk width, views): with c = ceil(r / 32) or c + 1 chunks of EQUAL width r / c the chunks are column VIEWS of
    the [N, r] matrix (row stride r: the *_ld entry points; an equal split keeps every view aligned to its rank's access width);
    otherwise c zero-padded copies of width ceil(r / c)."""
    c = -(-r // _MAXR)
    if os.environ.get("PSGD_WIDE_COPIES") != "1":             # (A/B switch of tools/r04_wide_rank_time.py: the round-3 chunk copies)
        for cc in (c, c + 1):
            if r % cc == 0 and r // cc <= _MAXR:
                return cc, r // cc, True
    return c, -(-r // c), False


def _ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _full_ok(*mats):
    """Ranks 33 .. 64 run on the WHOLE matrices (round 5: psgd_uvd_wide_*): contiguous [N, r], 16-byte aligned.  PSGD_WIDE_FULL=0: the
    column-chunk path (A/B runs)."""
    if os.environ.get("PSGD_WIDE_FULL") == "0" or os.environ.get("PSGD_WIDE_GRAM") == "0":
        return False
    r = mats[0].shape[1]
    return _MAXR < r <= 2 * _MAXR and all(m.dim() == 2 and m.is_contiguous() and m.data_ptr() % 16 == 0 for m in mats)


_wide_scratch = {}        # (device, stream, r) -> scratch of the psgd_uvd_wide_* entry points


class _Ctx:
    def __init__(self, U, workspace_fn, full=False):
        self.dev = U.device
        self.N, self.r = U.shape
        self.lib = _lib.load()
        self.st = torch.cuda.current_stream(self.dev).cuda_stream
        self.full = bool(full)
        if self.full:                                    # one "chunk": the matrix itself, through the rank 33 .. 64 kernels
            self.c, self.rc, self.views = 1, self.r, True
            key = (self.dev.index, self.st, self.r)
            n = int(self.lib.psgd_uvd_wide_scratch_bytes(self.N, self.r))
            if n < 0:
                _lib.check(n, "psgd_uvd_wide_scratch_bytes")
            scr = _wide_scratch.get(key)
            if scr is None or scr.numel() < n:
                scr = _wide_scratch[key] = torch.empty(n, dtype=torch.uint8, device=self.dev)
                while len(_wide_scratch) > 8:
                    _wide_scratch.pop(next(iter(_wide_scratch)))
            self.ws = scr
            return
        self.c, self.rc, self.views = _chunks(self.r)
        self.ws = workspace_fn(self.dev, self.N, self.rc)

    def split(self, M, copy=False):
        """column chunks [N, rc] of M [N, r]: views of M (no copy; in-place kernels then update M itself) when the rank splits
        evenly and M is row-major, else contiguous copies (the last one zero-padded; scatter() writes them back)."""
        if self.full and not copy:
            return [M]
        if self.views and not copy and M.dim() == 2 and M.stride(1) == 1 and self._view_ok(M):
            return [M[:, k * self.rc:(k + 1) * self.rc] for k in range(self.c)]
        out = []
        for k in range(self.c):
            lo, hi = k * self.rc, min((k + 1) * self.rc, self.r)
            # always a COPY in fresh (aligned) memory: a [1, rc] slice is "contiguous" as it stands, and callers update chunks in place
            if hi - lo < self.rc:
                ch = torch.zeros(self.N, self.rc, dtype=M.dtype, device=self.dev)
                ch[:, :hi - lo] = M[:, lo:hi]
            else:
                ch = torch.empty(self.N, self.rc, dtype=M.dtype, device=self.dev)
                ch.copy_(M[:, lo:hi])
            out.append(ch)
        return out

    def _view_ok(self, M):
        """alignment of the column views of M for the rank's access width (4, 2 or 1 floats)."""
        lv = 4 if self.rc % 4 == 0 else (2 if self.rc % 2 == 0 else 1)
        return M.stride(0) >= self.r and M.stride(0) % lv == 0 and M.data_ptr() % (4 * lv) == 0

    def scatter(self, M, chunks):
        for k, ch in enumerate(chunks):
            if ch.data_ptr() == M.data_ptr() + k * self.rc * M.element_size() and ch.stride(0) == M.stride(0):
                continue                                       # a view of M: already in place
            lo, hi = k * self.rc, min((k + 1) * self.rc, self.r)
            M[:, lo:hi] = ch[:, :hi - lo]

    def pad_vec(self, x):
        """r-vector (device) -> [c, rc] with zero padding."""
        out = torch.zeros(self.c * self.rc, dtype=x.dtype, device=self.dev)
        out[:self.r] = x.reshape(-1)
        return out.view(self.c, self.rc)

    def colsums(self, M, xs):
        S = torch.empty(len(xs), self.rc, dtype=torch.float64, device=self.dev)
        if self.full:
            _lib.check(self.lib.psgd_uvd_wide_colsums_f32(M.data_ptr(), _ptrs(xs), len(xs), S.data_ptr(), self.N, self.r,
                                                          self.ws.data_ptr(), self.ws.numel(), self.st), "psgd_uvd_wide_colsums_f32")
            return S
        _lib.check(self.lib.psgd_uvd_colsums_ld_f32(M.data_ptr(), M.stride(0), _ptrs(xs), len(xs), S.data_ptr(), self.N, self.rc,
                                                    self.ws.data_ptr(), self.ws.numel(), self.st), "psgd_uvd_colsums_ld_f32")
        return S

    def axpy(self, M, xs, S):
        """in place: x_j += M S_j  (S [k, rc] fp32)."""
        S = S.to(torch.float32).contiguous()
        p = _ptrs(xs)
        if self.full:
            _lib.check(self.lib.psgd_uvd_wide_axpy_cols_f32(M.data_ptr(), p, p, len(xs), S.data_ptr(), self.N, self.r, self.st),
                       "psgd_uvd_wide_axpy_cols_f32")
            return
        _lib.check(self.lib.psgd_uvd_axpy_cols_ld_f32(M.data_ptr(), M.stride(0), p, p, len(xs), S.data_ptr(), self.N, self.rc,
                                                      self.ws.data_ptr(), self.ws.numel(), self.st), "psgd_uvd_axpy_cols_ld_f32")

    def rank2(self, M, a, b, c1, c2):
        cc = torch.cat([c1.reshape(-1), c2.reshape(-1)]).to(torch.float32).contiguous()
        if self.full:
            _lib.check(self.lib.psgd_uvd_wide_rank2_update_f32(M.data_ptr(), a.data_ptr(), b.data_ptr(), cc.data_ptr(), self.N, self.r,
                                                               self.st), "psgd_uvd_wide_rank2_update_f32")
            return
        _lib.check(self.lib.psgd_uvd_rank2_update_ld_f32(M.data_ptr(), M.stride(0), a.data_ptr(), b.data_ptr(), cc.data_ptr(),
                                                         self.N, self.rc, self.ws.data_ptr(), self.ws.numel(), self.st),
                   "psgd_uvd_rank2_update_ld_f32")

    def gram_wide(self, U, V, d, v, h):
        """dense fp64 Gram [2r + 2, 2r + 2] of [U | V | d .* h | v ./ d], 32 < r <= 64, one sweep (psgd_uvd_gram_wide_f32)."""
        n = int(self.lib.psgd_uvd_gram_wide_scratch_bytes(self.N, self.r))
        if n < 0:
            _lib.check(n, "psgd_uvd_gram_wide_scratch_bytes")
        key = (self.dev.index, self.st, self.N, self.r)      # per stream: two streams must not share the fp64 partials
        scr = _gram_scratch.get(key)
        if scr is None or scr.numel() < n:
            scr = _gram_scratch[key] = torch.empty(n, dtype=torch.uint8, device=self.dev)
            while len(_gram_scratch) > 4:
                _gram_scratch.pop(next(iter(_gram_scratch)))
        G = torch.empty(2 * self.r + 2, 2 * self.r + 2, dtype=torch.float64, device=self.dev)
        _lib.check(self.lib.psgd_uvd_gram_wide_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(), h.data_ptr(), self.N, self.r,
                                                   G.data_ptr(), scr.data_ptr(), scr.numel(), self.st), "psgd_uvd_gram_wide_f32")
        return G

    # ---- Gram of [X | Y | t | w] for two chunks, decoded from the MFMA block layout of the workspace
    def gram_pair(self, X, Y, d, v, h):
        _lib.check(self.lib.psgd_uvd_update_sweep1_ld_f32(X.data_ptr(), X.stride(0), Y.data_ptr(), Y.stride(0), d.data_ptr(),
                                                          v.data_ptr(), h.data_ptr(), self.N, self.rc, self.ws.data_ptr(),
                                                          self.ws.numel(), self.st), "psgd_uvd_update_sweep1_ld_f32")
        off, cnt = _lib.ws_region(_lib.PSGD_WS_SUMS_F64, 11, self.N, self.rc)
        sums = self.ws[off:off + cnt * 8].view(torch.float64)
        return sums[self._gram_index()].clone()          # dense [2rc + 2, 2rc + 2]

    def _gram_index(self):
        if not hasattr(self, "_gidx"):
            nc = 2 * self.rc + 2
            nb = (nc + 15) // 16
            idx = torch.empty(nc, nc, dtype=torch.long)
            for a in range(nc):
                for b in range(nc):
                    lo, hi = (a, b) if a <= b else (b, a)
                    bi, bj, i, j = lo >> 4, hi >> 4, lo & 15, hi & 15
                    p = bi * nb - (bi * (bi - 1)) // 2 + (bj - bi)
                    idx[a, b] = p * 256 + (i & 3) * 64 + (((i >> 2) << 4) | j)
            self._gidx = idx.to(self.dev)
        return self._gidx


def _no_reduce(t, op):
    return t


def _update_native_ok(U, V, dv):
    """psgd_uvd_wide_update_f32: whole contiguous matrices, d contiguous and 16-byte aligned (updated in place).  PSGD_WIDE_UPDATE=0: the
    building-block route (A/B runs, and what a row-sharded update takes)."""
    if os.environ.get("PSGD_WIDE_UPDATE") == "0":
        return False
    return _full_ok(U, V) and _gram_wide_ok(U, V) and dv.is_contiguous() and dv.data_ptr() % 16 == 0


def _gram_wide_ok(U, V):
    """psgd_uvd_gram_wide_f32 takes contiguous [N, r] matrices (16-byte aligned when r % 4 == 0); PSGD_WIDE_GRAM=0: the pair-of-chunks
    route (A/B runs)."""
    if os.environ.get("PSGD_WIDE_GRAM") == "0":
        return False
    al = 16 if U.shape[1] % 4 == 0 else 4
    return U.is_contiguous() and V.is_contiguous() and U.data_ptr() % al == 0 and V.data_ptr() % al == 0


def precond_grad(U, V, d, g, workspace_fn, reduce=_no_reduce):
    """psgd.py:619-627 for r > 32.  g: a column vector, or a LIST of k contiguous [N] columns of a matrix g (returns [k, N])."""
    many = isinstance(g, (list, tuple))
    dv = d.reshape(-1)
    if reduce is _no_reduce and _full_ok(U, V):
        # ranks 33 .. 64 on one GPU: the three sweeps of the specialised apply on four columns at a time (psgd_uvd_wide_apply_cols_f32)
        cx = _Ctx(U, workspace_fn, full=True)
        cols = [x.reshape(-1).contiguous() for x in (g if many else [g])]
        outs = [torch.empty_like(x) for x in cols]
        dc = dv.contiguous()
        _lib.check(cx.lib.psgd_uvd_wide_apply_cols_f32(U.data_ptr(), V.data_ptr(), dc.data_ptr(), _ptrs(cols), _ptrs(outs), len(cols),
                                                       cx.N, cx.r, cx.ws.data_ptr(), cx.ws.numel(), cx.st), "psgd_uvd_wide_apply_cols_f32")
        return torch.stack(outs) if many else outs[0].reshape(g.shape)
    cx = _Ctx(U, workspace_fn, full=_full_ok(U, V))
    Uc, Vc = cx.split(U), cx.split(V)
    g1 = [(dv * x.reshape(-1)).contiguous() for x in (g if many else [g])]     # :625  t = d .* g
    s1 = reduce(torch.stack([cx.colsums(Vc[k], g1) for k in range(cx.c)]), "sum")        # V't, all chunks: one exchange
    for k in range(cx.c):                                                      # g1 = t + U (V't)          :544
        cx.axpy(Uc[k], g1, s1[k])
    out = [x.clone() for x in g1]
    s2 = reduce(torch.stack([cx.colsums(Uc[k], g1) for k in range(cx.c)]), "sum")        # U'g1
    for k in range(cx.c):                                                      # g1 + V (U'g1)             :626
        cx.axpy(Vc[k], out, s2[k])
    if many:
        return torch.stack([dv * x for x in out])
    return (dv * out[0]).reshape(g.shape)


def ipuvt_matvec(U, V, x, workspace_fn, reduce=_no_reduce):
    """psgd.py:540-544 for r > 32; x is [N], [N, 1] or [N, k]."""
    cx = _Ctx(U, workspace_fn, full=_full_ok(U, V))
    Uc, Vc = cx.split(U), cx.split(V)
    cols = [x.reshape(cx.N, -1)[:, j].contiguous() for j in range(x.reshape(cx.N, -1).shape[1])]
    outs = [c.clone() for c in cols]
    S = reduce(torch.stack([cx.colsums(Vc[k], cols) for k in range(cx.c)]), "sum")
    for k in range(cx.c):
        cx.axpy(Uc[k], outs, S[k])
    return torch.stack(outs, 1).reshape(x.shape) if x.dim() == 2 else outs[0].reshape(x.shape)


def _balance(U, V, reduce):
    """psgd.py:562-567"""
    mx = reduce(torch.stack([torch.max(torch.abs(U)), torch.max(torch.abs(V))]), "max")
    rho = torch.sqrt(mx[0] / mx[1])
    U.div_(rho)
    V.mul_(rho)


def _native_scratch(cx, fn_name):
    n = int(getattr(cx.lib, fn_name)(cx.N, cx.r))
    if n < 0:
        _lib.check(n, fn_name)
    key = (cx.dev.index, cx.st, cx.N, cx.r)
    scr = _update_scratch.get(key)
    if scr is None or scr.numel() < n:
        scr = _update_scratch[key] = torch.empty(n, dtype=torch.uint8, device=cx.dev)
        while len(_update_scratch) > 4:
            _update_scratch.pop(next(iter(_update_scratch)))
    return scr


def update_apply(U, V, d, v, h, g, step, tiny, balance, update_U, workspace_fn, reduce=_no_reduce):
    """update_precond_UVd_math_ followed by precond_grad_UVd_math on the updated state (psgd.py:732 -> :748) for r > 32.  Ranks 33 .. 64
    on one GPU with a column-vector g: psgd_uvd_wide_update_apply_f32 (three reads of U and V); otherwise the two calls.
    PSGD_WIDE_STEP=0: always the two calls (A/B runs)."""
    dv = d.reshape(-1)
    if (reduce is _no_reduce and os.environ.get("PSGD_WIDE_STEP") != "0" and not isinstance(g, (list, tuple)) and g.numel() == U.shape[0]
            and _update_native_ok(U, V, dv)):
        cx = _Ctx(U, workspace_fn, full=True)
        if balance:
            _balance(U, V, reduce)
        vc, hc, gc = v.reshape(-1).contiguous(), h.reshape(-1).contiguous(), g.reshape(-1).contiguous()
        out = torch.empty_like(gc)
        scr = _native_scratch(cx, "psgd_uvd_wide_update_apply_scratch_bytes")
        _lib.check(cx.lib.psgd_uvd_wide_update_apply_f32(U.data_ptr(), V.data_ptr(), dv.data_ptr(), vc.data_ptr(), hc.data_ptr(),
                                                         gc.data_ptr(), out.data_ptr(), cx.N, cx.r, float(step), float(tiny),
                                                         int(bool(update_U)), scr.data_ptr(), scr.numel(), cx.st),
                   "psgd_uvd_wide_update_apply_f32")
        return out.reshape(g.shape)
    if reduce is not _no_reduce and not isinstance(g, (list, tuple)) and g.numel() == U.shape[0] and os.environ.get("PSGD_WIDE_STEP") != "0":
        # row-sharded (round 6): the fused sequence on the building blocks -- TWO exchanges (the Gram; the 4r column sums of the
        # updated factors with max|nablaD| in one buffer), like the specialised ranks, instead of the four of update + apply
        return update(U, V, d, v, h, step, tiny, balance, update_U, workspace_fn, reduce, fused_g=g)
    update(U, V, d, v, h, step, tiny, balance, update_U, workspace_fn, reduce)
    return precond_grad(U, V, d, g, workspace_fn, reduce)


def update(U, V, d, v, h, step, tiny, balance, update_U, workspace_fn, reduce=_no_reduce, fused_g=None):
    """psgd.py:554-617 for r > 32 (in place on U or V, and d).
    fused_g (a column vector g): also precond_grad_UVd_math on the UPDATED state (psgd.py:732 -> :748), returned -- the sums the apply
    needs, [Unew | Vnew]' [d.*g, d.*g.*nablaD] with the OLD d, travel with max|nablaD| in ONE exchange (reduce(..., "summax"): the last
    entry is reduced by max), and the two r-vectors of the apply follow from them and the Gram (what k_fused_post does for r <= 64)."""
    cx = _Ctx(U, workspace_fn, full=_full_ok(U, V))
    r, c, rc, dev = cx.r, cx.c, cx.rc, cx.dev
    if balance:                                                                # :562-567
        _balance(U, V, reduce)
    dv, vv, hv = d.reshape(-1), v.reshape(-1), h.reshape(-1)
    if reduce is _no_reduce and _update_native_ok(U, V, dv):
        # ranks 33 .. 64 on one GPU: the four launches of psgd_uvd_wide_update_f32 (Gram, r x r algebra, sweep 2, d) -- no host step
        vc, hc = vv.contiguous(), hv.contiguous()
        scr = _native_scratch(cx, "psgd_uvd_wide_update_scratch_bytes")
        _lib.check(cx.lib.psgd_uvd_wide_update_f32(U.data_ptr(), V.data_ptr(), dv.data_ptr(), vc.data_ptr(), hc.data_ptr(), cx.N, cx.r,
                                                   float(step), float(tiny), int(bool(update_U)), scr.data_ptr(), scr.numel(), cx.st),
                   "psgd_uvd_wide_update_f32")
        return None
    Uc, Vc = cx.split(U), cx.split(V)
    # ---- Gram of W = [U | V | t | w], t = d.*h, w = v./d, block by block (pairs of chunks)
    R = c * rc
    f64 = dict(dtype=torch.float64, device=dev)
    UU, VV, VU = torch.zeros(R, R, **f64), torch.zeros(R, R, **f64), torch.zeros(R, R, **f64)
    Ut, Uw, Vt, Vw = (torch.zeros(R, **f64) for _ in range(4))
    items = [("U", k, Uc[k]) for k in range(c)] + [("V", k, Vc[k]) for k in range(c)]
    sl = lambda k: slice(k * rc, (k + 1) * rc)
    tt = tw = ww = None

    def put(kind_a, ka, kind_b, kb, blk):                 # blk = A'B  [rc, rc]
        if kind_a == "U" and kind_b == "U":
            UU[sl(ka), sl(kb)] = blk
            UU[sl(kb), sl(ka)] = blk.t()
        elif kind_a == "V" and kind_b == "V":
            VV[sl(ka), sl(kb)] = blk
            VV[sl(kb), sl(ka)] = blk.t()
        elif kind_a == "V":
            VU[sl(ka), sl(kb)] = blk
        else:
            VU[sl(kb), sl(ka)] = blk.t()

    if _MAXR < r <= 2 * _MAXR and _gram_wide_ok(U, V):
        # round 5: the whole Gram of [U | V | t | w] from ONE sweep over U and V (psgd_uvd_gram_wide_f32: the four waves of a workgroup
        # share a row tile and split the block pairs); the pair-of-chunks route below needs 2c - 1 passes
        G = reduce(cx.gram_wide(U, V, dv, vv, hv), "sum")                      # one exchange
        UU[:r, :r], VV[:r, :r], VU[:r, :r] = G[:r, :r], G[r:2 * r, r:2 * r], G[r:2 * r, :r]
        Ut[:r], Uw[:r], Vt[:r], Vw[:r] = G[:r, 2 * r], G[:r, 2 * r + 1], G[r:2 * r, 2 * r], G[r:2 * r, 2 * r + 1]
        tt, tw, ww = G[2 * r, 2 * r], G[2 * r, 2 * r + 1], G[2 * r + 1, 2 * r + 1]
    else:
        pairs = [(i, j) for i in range(len(items)) for j in range(i + 1, len(items))]
        Gs = reduce(torch.stack([cx.gram_pair(items[i][2], items[j][2], dv, vv, hv) for i, j in pairs]), "sum")   # one exchange
        for (i, j), G in zip(pairs, Gs):
            (ka_, ia, _), (kb_, ib, _) = items[i], items[j]
            put(ka_, ia, ka_, ia, G[:rc, :rc])
            put(kb_, ib, kb_, ib, G[rc:2 * rc, rc:2 * rc])
            put(ka_, ia, kb_, ib, G[:rc, rc:2 * rc])
            for kind, kk, lo in ((ka_, ia, 0), (kb_, ib, rc)):
                (Ut if kind == "U" else Vt)[sl(kk)] = G[lo:lo + rc, 2 * rc]
                (Uw if kind == "U" else Vw)[sl(kk)] = G[lo:lo + rc, 2 * rc + 1]
            tt, tw, ww = G[2 * rc, 2 * rc], G[2 * rc, 2 * rc + 1], G[2 * rc + 1, 2 * rc + 1]
    # ---- r x r algebra (fp64, on the device): psgd.py:574-579, :589-597 / :603-611
    K = torch.eye(R, **f64) + VU                                               # :575  (padded rows/columns: identity)
    s1 = Vt                                                                    # V't
    s2 = Ut + UU @ s1                                                          # U'Qh
    x1 = torch.linalg.solve(K.t(), Uw)                                         # :577 (adjoint)
    p2 = Vw - VV @ x1
    x2 = torch.linalg.solve(K, p2)                                             # :578
    cs1 = VU @ s1
    aa = tt + 2 * (s1 @ Ut) + s1 @ (UU @ s1)
    bb = ww - 2 * (x1 @ Vw) + x1 @ (VV @ x1)
    ab = tw - x1 @ Vt + s1 @ Uw - x1 @ cs1
    if update_U:
        e1, e2, Mm = Vt + cs1, p2, VV                                          # atV, btV
    else:
        e1, e2, Mm = s2, Uw - VU.t() @ x1, UU                                  # atU, btU
    nrm = torch.sqrt(torch.abs(aa * (e1 @ (Mm @ e1)) + bb * (e2 @ (Mm @ e2)) - 2 * ab * (e1 @ (Mm @ e2))))
    mu = (step / (nrm + tiny)).to(torch.float32)                               # :597 / :611
    if update_U:
        c1, c2 = e1 @ K, e2 @ K                                                # atV K, btV K          :600-601
    else:
        c1, c2 = e1, e2
    # ---- row-local part: a = Qh, b = invQtv, Ph, invPv, nablaD (psgd.py:569-581); two vectors per sweep of a chunk
    t = (dv * hv).contiguous()
    w = (vv / dv).contiguous()
    a, q = t.clone(), torch.zeros_like(t)                  # a = t + U s1 ;  q = -U x2
    b, p = w.clone(), torch.zeros_like(t)                  # b = w - V x1 ;  p = V s2
    for k in range(c):
        cx.axpy(Uc[k], [a, q], torch.stack([s1[sl(k)], -x2[sl(k)]]))
        cx.axpy(Vc[k], [b, p], torch.stack([-x1[sl(k)], s2[sl(k)]]))
    Ph = dv * (a + p)                                                          # :570
    invPv = (b + q) / dv                                                       # :578-579
    nabla = Ph * hv - vv * invPv                                               # :581
    if fused_g is None:
        mud = step / (reduce(torch.max(torch.abs(nabla)).reshape(1), "max")[0] + tiny)    # :582
        dv.sub_((mud * dv) * nabla)                                            # :584 (d before U / V; a, b keep the old d)
    c1f, c2f = (mu * c1.to(torch.float32)), (mu * c2.to(torch.float32))
    if update_U:                                                               # :600-601
        for k in range(c):
            cx.rank2(Uc[k], a, b, c1f[sl(k)], c2f[sl(k)])
        cx.scatter(U, Uc)
    else:                                                                      # :614-615
        al, be = a.clone(), b.clone()
        for k in range(c):
            cx.axpy(Vc[k], [al, be], torch.stack([c1[sl(k)], c2[sl(k)]]))
        for k in range(c):
            cx.rank2(Vc[k], al, be, c1f[sl(k)], c2f[sl(k)])
        cx.scatter(V, Vc)
    if fused_g is None:
        return None
    # ---- the apply on the updated state from ONE more exchange (the factor update above does not need mu_d, so it went first)
    gv = fused_g.reshape(-1)
    tg = (dv * gv).contiguous()                                                # d .* g            (old d)
    tn = (tg * nabla).contiguous()                                             # d .* g .* nablaD
    sums = torch.stack([torch.stack([cx.colsums(Uc[k], [tg, tn]), cx.colsums(Vc[k], [tg, tn])]) for k in range(c)])   # [c, U|V, tg|tn, rc]
    packed = reduce(torch.cat([sums.reshape(-1), torch.max(torch.abs(nabla)).to(torch.float64).reshape(1)]), "summax")
    S = packed[:-1].view(c, 2, 2, rc)
    pU, pV, qU, qV = (S[:, i, j, :].reshape(-1) for i, j in ((0, 0), (1, 0), (0, 1), (1, 1)))
    mud = step / (packed[-1].to(torch.float32) + tiny)                         # :582 (the maximum is an fp32 value)
    Anew = UU
    if update_U:                                           # Unew = U - a c1f' + b c2f':  Unew'Unew from U'U, U'a = s2, U'b, a'a, a'b, b'b
        f1, f2 = c1f.to(torch.float64), c2f.to(torch.float64)
        Ua, Ub = s2, Uw - VU.t() @ x1
        o = torch.outer
        Anew = UU - (o(Ua, f1) + o(f1, Ua)) + (o(Ub, f2) + o(f2, Ub)) + aa * o(f1, f1) - ab * (o(f1, f2) + o(f2, f1)) + bb * o(f2, f2)
    s1n = pV - mud.to(torch.float64) * qV                                      # Vnew'(dnew .* g)
    s2n = pU - mud.to(torch.float64) * qU + Anew @ s1n                         # Unew'(dnew .* g + Unew s1')
    dv.sub_((mud * dv) * nabla)                                                # :584
    acc = (dv * gv).contiguous()
    for k in range(c):
        cx.axpy(Uc[k], [acc], s1n[sl(k)].reshape(1, -1))
        cx.axpy(Vc[k], [acc], s2n[sl(k)].reshape(1, -1))
    return (dv * acc).reshape(fused_g.shape)                                   # :626
