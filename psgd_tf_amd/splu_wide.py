"""Sparse-LU preconditioner of rank r > 64 (the reference has no rank limit, psgd.py:396-524); ranks 33 .. 64 only under
PSGD_SPLU_CHUNKS=1 (A/B runs): since round 5 the native tail kernels cover r = 1..64.

This route builds on the rank <= 32 building blocks of the wide UVd path.  A wider preconditioner runs on column chunks
of its two tall blocks -- L2 [N - r, r] and U2' [N - r, r], every chunk an [N - r, rc] matrix with rc <= 32: column VIEWS of L2
when the rank splits evenly (round 4: the `*_ld` entry points), contiguous copies otherwise (the last one zero-padded) and for
U2' (its rows lie along the columns of U12: one transpose per chunk) -- through the same three building blocks of the C ABI as the
wide-rank UVd path (uvd_wide.py):

    S = M' [x ..]                psgd_uvd_colsums_f32       U2 x2, L2' x2                    (psgd.py:430,437,440,442,452)
    out = x + M S                psgd_uvd_axpy_cols_f32     L2 s + ., U2' s + .              (psgd.py:434,446,449)
    M <- M - (a c1' - b c2')     psgd_uvd_rank2_update_f32  the rank-2 parts of :464 and :477

What is left to torch has no (N - r) x r extent except the chunk copies themselves and the row scalings of :464 / :477
(one elementwise pass per chunk): N-vector arithmetic, the r x r corner (L1, U1, four triangular solves, the corner
gradients) in fp64, and the maxima of :459-461 / :472-474 (column blocks of the rank-2 gradient, never the whole N x r
matrix).  Everything stays on the device; nothing synchronises.  Same formulas and order of operations as psgd.py:396-524.
Row-sharded use through the `reduce` hook (psgd_tf_amd/sharded.py routes r > 32 here, round 4).
"""
import torch

from . import uvd_wide as _w

_f64 = torch.float64


def _no_reduce(t, op):
    return t


class _Blocks:
    """reduce(tensor, "sum" | "max"): applied to everything that is a sum or a maximum over the TAIL rows (row-sharded use,
    psgd_tf_amd/sharded.py: every rank holds the r x r corner and its rows of the tail; default: one GPU holds all rows).
    L2_work: the [N - r, r] block the chunks of L2 are taken from -- L12[r:] itself for the apply (column VIEWS when the rank
    splits evenly: nothing is copied), the tail rows of the OUTPUT L12 for the update (already holding the balanced L2: the
    in-place kernels then write the result where it belongs and the inputs stay untouched)."""

    def __init__(self, L12, l3, U12, u3, workspace_fn, L2_work=None, reduce=_no_reduce):
        self.reduce = reduce
        self.N, self.r = L12.shape
        self.n2 = self.N - self.r
        r = self.r
        self.dev = L12.device
        self.L1, self.U1 = L12[:r].to(_f64), U12[:, :r].to(_f64)
        self.l3, self.u3 = l3.reshape(-1), u3.reshape(-1)
        if self.n2 > 0:
            L2 = L12[r:] if L2_work is None else L2_work
            # rows = N - r, columns = r -> c chunks of rc; ranks 33 .. 64 (round 5): ONE "chunk", the block itself, through the
            # whole-matrix kernels of the wide UVd path (psgd_uvd_wide_*) when L2 is contiguous and 16-byte aligned
            self.cx = _w._Ctx(L2, workspace_fn, full=reduce is _no_reduce and _w._full_ok(L2))
            self.Lc = self.cx.split(L2)
            self.L2 = L2
            rc = self.cx.rc
            self.Uc = []
            for k in range(self.cx.c):                               # chunks of U2' (rows of U12 are strided by N: one transpose each)
                lo, hi = k * rc, min((k + 1) * rc, r)
                ch = torch.zeros(self.n2, rc, dtype=L12.dtype, device=self.dev)
                ch[:, :hi - lo] = U12[lo:hi, r:].t()
                self.Uc.append(ch)

    def _sl(self, k):
        rc = self.cx.rc
        return slice(k * rc, min((k + 1) * rc, self.r))

    def t_times(self, chunks, x):
        """[chunks]' x  (r-vector, fp64) for an (N - r)-vector x."""
        if self.n2 == 0:
            return self.reduce(torch.zeros(self.r, dtype=_f64, device=self.dev), "sum")
        S = torch.stack([self.cx.colsums(ch, [x])[0] for ch in chunks])          # [c, rc]; only the last chunk is padded
        return self.reduce(S.reshape(-1)[:self.r].contiguous(), "sum")

    def times_plus(self, chunks, x, s):
        """x + [chunks] s  (new (N - r)-vector) for an r-vector s."""
        out = x.clone()
        if self.n2 == 0:
            return out
        sp = self.cx.pad_vec(s)
        for k, ch in enumerate(chunks):
            self.cx.axpy(ch, [out], sp[k:k + 1])
        return out


def _flat(x):
    return x.reshape(-1)


def precond_grad(L12, l3, U12, u3, g, workspace_fn, reduce=_no_reduce):
    """psgd.py:499-516 for r > 32; g is the flat gradient [N]; returns the flat preconditioned gradient.  (2 exchanges when
    row-sharded: U2 g2, L2' Qg2.)"""
    b = _Blocks(L12, l3, U12, u3, workspace_fn, reduce=reduce)
    r = b.r
    g1, g2 = g[:r].to(_f64), g[r:].contiguous()
    Ug1 = b.U1 @ g1 + b.t_times(b.Uc if b.n2 else None, g2)                     # :506
    Qg1 = b.L1 @ Ug1                                                            # :509
    Qg2 = b.times_plus(b.Lc if b.n2 else None, b.l3 * (b.u3 * g2), Ug1)         # :507,510
    LtQg1 = b.L1.t() @ Qg1 + b.t_times(b.Lc if b.n2 else None, Qg2)             # :512
    pre1 = b.U1.t() @ LtQg1                                                     # :515
    pre2 = b.times_plus(b.Uc if b.n2 else None, b.u3 * (b.l3 * Qg2), LtQg1)     # :513,516
    return torch.cat([pre1.to(g.dtype), pre2])


def _max_abs_rank2(a, p, bvec, q, block=8):
    """max |a_i p_j - b_i q_j| over all i, j without forming the N x r matrix: column blocks of `block` columns."""
    m = None
    pf, qf = p.to(a.dtype), q.to(a.dtype)
    for j0 in range(0, pf.numel(), block):
        t = torch.max(torch.abs(a[:, None] * pf[None, j0:j0 + block] - bvec[:, None] * qf[None, j0:j0 + block]))
        m = t if m is None else torch.maximum(m, t)           # (torch.max / torch.maximum propagate NaN, as tf.reduce_max)
    return m


def update(L12, l3, U12, u3, dx, dg, step, tiny, workspace_fn, reduce=_no_reduce):
    """psgd.py:396-480 for r > 32; dx, dg flat [N]; returns (L12_new, l3_new, U12_new, u3_new).  (Row-sharded: 6 exchanges --
    the two signed maxima of :411-412 in one, four column-sum vectors, the three maxima of each gradient in one each.)"""
    N, r = L12.shape
    dev, dt = L12.device, L12.dtype
    l3f, u3f = l3.reshape(-1), u3.reshape(-1)
    ninf = torch.tensor(float("-inf"), dtype=dt, device=dev)
    mlu = reduce(torch.stack([torch.max(l3f) if l3f.numel() else ninf, torch.max(u3f) if u3f.numel() else ninf]), "max")
    max_l = torch.maximum(torch.max(torch.diagonal(L12[:r])), mlu[0])                                      # :411 (signed)
    max_u = torch.maximum(torch.max(torch.diagonal(U12[:, :r])), mlu[1])                                   # :412
    rho = torch.sqrt(max_l / max_u)                                                                        # :413
    L12n = torch.empty_like(L12)
    if N > r:
        torch.div(L12[r:], rho, out=L12n[r:])                                                              # :414 (L2 / rho, in the output)
    b = _Blocks(L12, l3, U12, u3, workspace_fn, L2_work=L12n[r:] if N > r else None, reduce=reduce)
    rho64 = rho.to(_f64)
    b.L1, b.U1 = b.L1 / rho64, b.U1 * rho64                                                                # :414-417
    b.l3, b.u3 = l3f / rho, u3f * rho
    if b.n2:
        for ch in b.Uc:
            ch.mul_(rho)
    Lc, Uc = (b.Lc, b.Uc) if b.n2 else (None, None)
    x1, x2 = dx[:r].to(_f64), dx[r:].contiguous()
    g1, g2 = dg[:r].to(_f64), dg[r:].contiguous()
    tri = torch.linalg.solve_triangular

    Ug1 = b.U1 @ g1 + b.t_times(Uc, g2)                                                                    # :430
    Qg1 = b.L1 @ Ug1                                                                                       # :433
    Qg2 = b.times_plus(Lc, b.l3 * (b.u3 * g2), Ug1)                                                        # :431,434
    iUtx1 = tri(b.U1.t(), x1[:, None], upper=False)[:, 0]                                                  # :436 (adjoint)
    iUtx2 = b.times_plus(Uc, x2, -iUtx1) / b.u3                                                            # :437
    iQtx2 = iUtx2 / b.l3                                                                                   # :439
    iQtx1 = tri(b.L1.t(), (iUtx1 - b.t_times(Lc, iQtx2))[:, None], upper=True)[:, 0]                        # :440 (adjoint)
    LtQg1 = b.L1.t() @ Qg1 + b.t_times(Lc, Qg2)                                                            # :442
    Pg1 = b.U1.t() @ LtQg1                                                                                 # :445
    Pg2 = b.times_plus(Uc, b.u3 * (b.l3 * Qg2), LtQg1)                                                     # :443,446
    iLiQtx1 = tri(b.L1, iQtx1[:, None], upper=False)[:, 0]                                                 # :448
    iLiQtx2 = b.times_plus(Lc, iQtx2, -iLiQtx1) / b.l3                                                     # :449
    iPx2 = iLiQtx2 / b.u3                                                                                  # :451
    iPx1 = tri(b.U1, (iLiQtx1 - b.t_times(Uc, iPx2))[:, None], upper=True)[:, 0]                            # :452

    # ---- update L (:455-465)
    grad1 = torch.tril(torch.outer(Qg1, Qg1) - torch.outer(iQtx1, iQtx1))
    grad3 = Qg2 * Qg2 - iQtx2 * iQtx2
    mx = torch.max(torch.abs(grad1)).to(dt)
    mt = torch.zeros((), dtype=dt, device=dev)                      # the tail's part of the maximum (0: none on this rank)
    if b.n2:
        mt = torch.maximum(_max_abs_rank2(Qg2, Qg1, iQtx2, iQtx1), torch.max(torch.abs(grad3)))
    mx = torch.maximum(mx, reduce(mt.reshape(1), "max")[0])
    step0 = step / (mx + tiny)                                                                             # :462
    s64 = step0.to(_f64)
    newL1 = b.L1 - (s64 * grad1) @ b.L1                                                                    # :463
    L12n[:r] = newL1.to(dt)
    if b.n2:
        c1, c2 = s64 * (Qg1 @ b.L1), s64 * (iQtx1 @ b.L1)             # (grad2 L1) = Qg2 (Qg1' L1) - iQtx2 (iQtx1' L1)
        c1p, c2p = b.cx.pad_vec(c1), b.cx.pad_vec(c2)
        scale = (1.0 - step0 * grad3)[:, None]
        for k, ch in enumerate(Lc):                                                                        # :464
            ch.mul_(scale)
            b.cx.rank2(ch, Qg2, iQtx2, c1p[k], c2p[k])
        b.cx.scatter(b.L2, Lc)                                      # (copies only: views of L12n[r:] are in place already)
    l3n = (b.l3 - step0 * grad3 * b.l3).reshape(l3.shape)                                                  # :465

    # ---- update U (:468-478)
    grad1 = torch.triu(torch.outer(Pg1, g1) - torch.outer(x1, iPx1))
    grad3 = Pg2 * g2 - x2 * iPx2
    mx = torch.max(torch.abs(grad1)).to(dt)
    mt = torch.zeros((), dtype=dt, device=dev)
    if b.n2:
        mt = torch.maximum(_max_abs_rank2(g2, Pg1, iPx2, x1), torch.max(torch.abs(grad3)))      # |Pg1_j dg2_i - dx1_j iPx2_i|
    mx = torch.maximum(mx, reduce(mt.reshape(1), "max")[0])
    step0 = step / (mx + tiny)                                                                             # :475
    s64 = step0.to(_f64)
    newU1 = b.U1 - b.U1 @ (s64 * grad1)                                                                    # :476
    U12n = torch.empty_like(U12)
    U12n[:, :r] = newU1.to(dt)
    if b.n2:
        c1, c2 = s64 * (b.U1 @ Pg1), s64 * (b.U1 @ x1)                # U1 grad2 = (U1 Pg1) dg2' - (U1 dx1) iPx2'
        c1p, c2p = b.cx.pad_vec(c1), b.cx.pad_vec(c2)
        scale = (1.0 - step0 * grad3)[:, None]
        for k, ch in enumerate(Uc):                                                                        # :477 (transposed chunks)
            ch.mul_(scale)
            b.cx.rank2(ch, g2, iPx2, c1p[k], c2p[k])
            sl = b._sl(k)
            U12n[sl, r:] = ch[:, :sl.stop - sl.start].t()
    u3n = (b.u3 - step0 * grad3 * b.u3).reshape(u3.shape)                                                  # :478
    return L12n, l3n, U12n, u3n
