"""CPU stage backend for tests/test_sharded_cpu.py.  TEST INFRASTRUCTURE ONLY.

Implements the stage interface of psgd_tf_amd.sharded.HipStages with NumPy, so that the
multi-rank choreography (which buffer is all-reduced when, SUM vs MAX, what each rank
recomputes redundantly) can run under gloo on a machine without a GPU.  The stage math restates
the fused plan of the HIP kernels: Gram of [U V t w] -> r x r algebra ->
row-local update; the test compares its sharded result with the reference-order oracle.
"""
import numpy as np
import torch


class _GatherFold:
    """The exchange half of the stage interface (send / gather_buf / fold): every rank's fp64 send region is
    all-gathered, then folded in rank order -- entries [0, nsum) by +, the rest by max (the product's
    psgd_*_fold_gathered_f64).  Subclasses say how a stage's send region is laid out (_send_layout)."""

    def send(self, stage):
        sums, maxes = self._send_layout(stage)
        parts = ([sums] if sums is not None else []) + ([maxes.to(torch.float64)] if maxes is not None else [])
        return torch.cat(parts).contiguous()

    def gather_buf(self, stage, world):
        return torch.empty(world * self.send(stage).numel(), dtype=torch.float64)

    def fold(self, stage, gathered, world):
        sums, maxes = self._send_layout(stage)
        nsum = sums.numel() if sums is not None else 0
        g = gathered.view(world, -1)
        acc = g[0].clone()
        for k in range(1, world):                      # rank order
            acc[:nsum] += g[k][:nsum]
            acc[nsum:] = torch.maximum(acc[nsum:], g[k][nsum:])
        if sums is not None:
            sums[:] = acc[:nsum]
        if maxes is not None:
            maxes[:] = acc[nsum:].to(torch.float32)


class NumpyStages(_GatherFold):
    def _send_layout(self, stage):
        if stage == 10:
            return None, self._max[10]
        if stage == 12:
            return None, self._max[12]
        if stage == 13:
            return self._sums[13], self._max[12]       # [pU | pV | qU | qV | max] in one region
        return self._sums[stage], None

    def __init__(self, r, dtype=np.float64):
        self.r = r
        self.dtype = dtype
        nc = 2 * r + 2
        self._sums = {1: torch.zeros(r, dtype=torch.float64), 2: torch.zeros(r, dtype=torch.float64),
                      11: torch.zeros(nc * nc, dtype=torch.float64), 13: torch.zeros(4 * r, dtype=torch.float64)}
        self._max = {10: torch.zeros(2, dtype=torch.float32), 12: torch.zeros(1, dtype=torch.float32)}
        self.nabla = None

    def sums(self, stage):
        return self._sums[stage]

    def maxbuf(self, stage):
        return self._max[stage]

    @staticmethod
    def _np(t):
        return t.numpy() if isinstance(t, torch.Tensor) else t

    # ---- apply (psgd.py:619-627)
    def apply_sweep1(self, V, d, g):
        V, d, g = map(self._np, (V, d, g))
        self._sums[1][:] = torch.from_numpy((V.T @ (d * g)).astype(np.float64).ravel())

    def apply_sweep2(self, U, d, g):
        U, d, g = map(self._np, (U, d, g))
        s1 = self._sums[1].numpy().astype(self.dtype).reshape(-1, 1)
        g1 = d * g + U @ s1
        self._sums[2][:] = torch.from_numpy((U.T @ g1).astype(np.float64).ravel())

    def apply_sweep3(self, U, V, d, g):
        Un, Vn, dn, gn = map(self._np, (U, V, d, g))
        s1 = self._sums[1].numpy().astype(self.dtype).reshape(-1, 1)
        s2 = self._sums[2].numpy().astype(self.dtype).reshape(-1, 1)
        g1 = dn * gn + Un @ s1
        return torch.from_numpy(dn * (g1 + Vn @ s2))

    # ---- update (psgd.py:554-617)
    def balance_max(self, U, V):
        self._max[10][:] = torch.tensor([np.max(np.abs(self._np(U))), np.max(np.abs(self._np(V)))], dtype=torch.float32)

    def balance_scale(self, U, V):
        m = self._max[10].numpy().astype(np.float64)
        rho = np.sqrt(m[0] / m[1])
        Un, Vn = self._np(U), self._np(V)
        Un /= rho
        Vn *= rho

    def update_sweep1(self, U, V, d, v, h):
        U, V, d, v, h = map(self._np, (U, V, d, v, h))
        W = np.concatenate([U, V, d * h, v / d], axis=1).astype(np.float64)
        self._sums[11][:] = torch.from_numpy((W.T @ W).ravel())

    def update_sweep2(self, U, V, d, v, h, step, tiny, update_U):
        r = self.r
        Un, Vn, dn, vn, hn = map(self._np, (U, V, d, v, h))
        G = self._sums[11].numpy().reshape(2 * r + 2, 2 * r + 2)
        A, B, C = G[:r, :r], G[r:2 * r, r:2 * r], G[r:2 * r, :r]           # U'U, V'V, V'U
        ut, uw, vt, vw = G[:r, 2 * r], G[:r, 2 * r + 1], G[r:2 * r, 2 * r], G[r:2 * r, 2 * r + 1]
        tt, tw, ww = G[2 * r, 2 * r], G[2 * r, 2 * r + 1], G[2 * r + 1, 2 * r + 1]
        K = np.eye(r) + C
        s1 = vt
        s2 = ut + A @ s1
        x1 = np.linalg.solve(K.T, uw)
        p2 = vw - B @ x1
        x2 = np.linalg.solve(K, p2)
        aa = tt + 2 * s1 @ ut + s1 @ A @ s1
        bb = ww - 2 * x1 @ vw + x1 @ B @ x1
        ab = tw - x1 @ vt + s1 @ uw - x1 @ (C @ s1)
        if update_U:
            e1, e2, Mm = vt + C @ s1, p2, B
        else:
            e1, e2, Mm = s2, uw - C.T @ x1, A
        nrm = np.sqrt(abs(aa * (e1 @ Mm @ e1) + bb * (e2 @ Mm @ e2) - 2 * ab * (e1 @ Mm @ e2)))
        mu = step / (nrm + tiny)
        c = lambda x: x.astype(self.dtype).reshape(-1, 1)
        t, w = dn * hn, vn / dn
        a = t + Un @ c(s1)
        b = w - Vn @ c(x1)
        Ph = dn * (a + Vn @ c(s2))
        invPv = (b - Un @ c(x2)) / dn
        self.nabla = Ph * hn - vn * invPv
        self._max[12][:] = torch.tensor([np.max(np.abs(self.nabla))], dtype=torch.float32)
        self._upd = dict(s1=s1, x1=x1, mu=mu, aa=aa, ab=ab, bb=bb, Ua=s2, Ub=uw - C.T @ x1, A=A.copy())
        if update_U:
            c1, c2 = e1 @ K, e2 @ K
            self._upd.update(c1=c1, c2=c2)
            Un -= mu * (a @ c(c1).T - b @ c(c2).T)
        else:
            al, be = a + Vn @ c(e1), b + Vn @ c(e2)
            Vn -= mu * (al @ c(e1).T - be @ c(e2).T)

    def update_sweep2_fused(self, U, V, d, v, h, g, step, tiny, update_U):
        """sweep 2 + the four extra column reductions of the fused update->apply:
        [Unew | Vnew]' [d.*g, d.*g.*nablaD] with the OLD d."""
        dn, gn = self._np(d), self._np(g)
        d_old = dn.copy()
        self.update_sweep2(U, V, d, v, h, step, tiny, update_U)
        Un, Vn = self._np(U), self._np(V)                  # one of them is the NEW factor
        tg, tn = d_old * gn, d_old * gn * self.nabla
        sums = np.concatenate([Un.T @ tg, Vn.T @ tg, Un.T @ tn, Vn.T @ tn])
        self._sums[13][:] = torch.from_numpy(sums.astype(np.float64).ravel())

    def fused_post(self, step, tiny, update_U):
        """r x r algebra on the exchanged sums (restates k_fused_post): s1' = pV - mu_d qV,
        s2' = pU - mu_d qU + (Unew'Unew) s1', Unew'Unew = U'U + the rank-2 correction of psgd.py:600-601."""
        r, u = self.r, self._upd
        pq = self._sums[13].numpy()
        pU, pV, qU, qV = pq[:r], pq[r:2 * r], pq[2 * r:3 * r], pq[3 * r:]
        mud = step / (float(self._max[12][0]) + tiny)
        A = u["A"]
        if update_U:
            mu, c1, c2, Ua, Ub = u["mu"], u["c1"], u["c2"], u["Ua"], u["Ub"]
            o = np.outer
            A = A - mu * (o(Ua, c1) + o(c1, Ua) - o(Ub, c2) - o(c2, Ub)) \
                + mu * mu * (u["aa"] * o(c1, c1) - u["ab"] * (o(c1, c2) + o(c2, c1)) + u["bb"] * o(c2, c2))
        s1n = pV - mud * qV
        s2n = pU - mud * qU + A @ s1n
        self._sums[1][:] = torch.from_numpy(s1n)
        self._sums[2][:] = torch.from_numpy(s2n)

    def fused_final(self, U, V, d, g, step, tiny):
        """last sweep: d update (psgd.py:584), then out = d (d g + U s1' + V s2')."""
        self.update_sweep3(d, step, tiny)
        Un, Vn, dn, gn = map(self._np, (U, V, d, g))
        s1 = self._sums[1].numpy().astype(self.dtype).reshape(-1, 1)
        s2 = self._sums[2].numpy().astype(self.dtype).reshape(-1, 1)
        return torch.from_numpy(dn * (dn * gn + Un @ s1 + Vn @ s2))

    def update_sweep3(self, d, step, tiny):
        dn = self._np(d)
        mu = step / (float(self._max[12][0]) + tiny)
        dn -= mu * dn * self.nabla


class NumpySpluStages(_GatherFold):
    """CPU stage backend for the sharded sparse LU (interface of psgd_tf_amd.sharded.HipSpluStages).  fp64 NumPy
    restatement of the staged plan of psgd_splu.hip: reductions over the tail rows -> r x r corner algebra ->
    row-local work; the test compares its sharded result with the reference-order oracle."""

    def _send_layout(self, stage):
        return (self._sums[3], self._max) if stage == 3 else (self._sums[stage], None)   # stage 3: [r sums | 4 maxima]

    def __init__(self, r):
        self.r = r
        self._sums = {1: torch.zeros(r, dtype=torch.float64), 2: torch.zeros(2 * r, dtype=torch.float64),
                      3: torch.zeros(r, dtype=torch.float64)}
        self._max = torch.zeros(4, dtype=torch.float32)

    def sums(self, stage):
        return self._sums[stage]

    def maxbuf(self):
        return self._max

    def _blocks(self, L12, U12):
        r = self.r
        L12, U12 = L12.numpy(), U12.numpy()
        return L12[:r], L12[r:], U12[:, :r], U12[:, r:]

    def stage1(self, U12, x):
        r = self.r
        self._sums[1][:] = torch.from_numpy((U12.numpy()[:, r:] @ x.numpy()[r:]).ravel())

    def apply_stage2(self, L12, l3, U12, u3, g):
        r = self.r
        L1, L2, U1, U2 = self._blocks(L12, U12)
        g = g.numpy()
        Ug1 = U1 @ g[:r] + self._sums[1].numpy().reshape(-1, 1)                     # psgd.py:506
        self.Qg1 = L1 @ Ug1                                                          # :509
        self.Qg2 = L2 @ Ug1 + l3.numpy() * (u3.numpy() * g[r:])                      # :507,510
        self._sums[2][:r] = torch.from_numpy((L2.T @ self.Qg2).ravel())

    def apply_stage3(self, L12, l3, U12, u3):
        r = self.r
        L1, L2, U1, U2 = self._blocks(L12, U12)
        Lt1 = L1.T @ self.Qg1 + self._sums[2].numpy()[:r].reshape(-1, 1)             # :512
        out = np.concatenate([U1.T @ Lt1, U2.T @ Lt1 + u3.numpy() * (l3.numpy() * self.Qg2)], 0)   # :513-516
        return torch.from_numpy(out)

    def update_stage2(self, L12, l3, U12, u3, dx, dg):
        r = self.r
        L1, L2, U1, U2 = self._blocks(L12, U12)
        x, g, l, u = dx.numpy(), dg.numpy(), l3.numpy(), u3.numpy()
        Ug1 = U1 @ g[:r] + self._sums[1].numpy().reshape(-1, 1)                      # :430
        self.Qg1 = L1 @ Ug1                                                          # :433
        self.iUtx1 = np.linalg.solve(np.triu(U1).T, x[:r])                           # :436
        self.Qg2 = L2 @ Ug1 + l * (u * g[r:])                                        # :431,434
        self.iQtx2 = (x[r:] - U2.T @ self.iUtx1) / u / l                             # :437,439
        self._sums[2][:] = torch.from_numpy(np.concatenate([L2.T @ self.Qg2, L2.T @ self.iQtx2]).ravel())

    def update_stage3(self, L12, l3, U12, u3, dx, dg):
        r = self.r
        L1, L2, U1, U2 = self._blocks(L12, U12)
        x, g, l, u = dx.numpy(), dg.numpy(), l3.numpy(), u3.numpy()
        sB = self._sums[2].numpy()
        self.iQtx1 = np.linalg.solve(np.tril(L1).T, self.iUtx1 - sB[r:].reshape(-1, 1))     # :440
        self.LtQg1 = L1.T @ self.Qg1 + sB[:r].reshape(-1, 1)                          # :442
        self.Pg1 = U1.T @ self.LtQg1                                                  # :445
        self.iLiQtx1 = np.linalg.solve(np.tril(L1), self.iQtx1)                       # :448
        self.Pg2 = U2.T @ self.LtQg1 + u * (l * self.Qg2)                             # :443,446
        self.iPx2 = (self.iQtx2 - L2 @ self.iLiQtx1) / l / u                          # :449,451
        self._sums[3][:] = torch.from_numpy((U2 @ self.iPx2).ravel())
        n2 = L2.shape[0]
        gL2, gL3 = self.Qg2 @ self.Qg1.T - self.iQtx2 @ self.iQtx1.T, self.Qg2 ** 2 - self.iQtx2 ** 2
        gU2, gU3 = self.Pg1 @ g[r:].T - x[:r] @ self.iPx2.T, self.Pg2 * g[r:] - x[r:] * self.iPx2
        mx = lambda a: float(np.max(np.abs(a))) if a.size else 0.0
        self._max[:] = torch.tensor([max(mx(gL2), mx(gL3)), max(mx(gU2), mx(gU3)),
                                     float(np.max(l)) if n2 else -np.inf, float(np.max(u)) if n2 else -np.inf],
                                    dtype=torch.float32)

    def update_stage4(self, L12, l3, U12, u3, dx, dg, step, tiny, has_tail):
        r = self.r
        L1, L2, U1, U2 = self._blocks(L12, U12)
        x, g, l, u = dx.numpy(), dg.numpy(), l3.numpy(), u3.numpy()
        m = self._max.numpy().astype(np.float64)
        iPx1 = np.linalg.solve(np.triu(U1), self.iLiQtx1 - self._sums[3].numpy().reshape(-1, 1))   # :452
        max_l = max(np.max(np.diag(L1)), m[2]) if has_tail else np.max(np.diag(L1))   # :411-413
        max_u = max(np.max(np.diag(U1)), m[3]) if has_tail else np.max(np.diag(U1))
        rho = np.sqrt(max_l / max_u)
        L1s, L2s, l3s, U1s, U2s, u3s = L1 / rho, L2 / rho, l / rho, rho * U1, rho * U2, rho * u
        gL1 = np.tril(self.Qg1 @ self.Qg1.T - self.iQtx1 @ self.iQtx1.T)               # :455-458
        gL2, gL3 = self.Qg2 @ self.Qg1.T - self.iQtx2 @ self.iQtx1.T, self.Qg2 ** 2 - self.iQtx2 ** 2
        s0 = step / (max(np.max(np.abs(gL1)), m[0] if has_tail else 0.0) + tiny)       # :459-462
        nL = np.concatenate([L1s - s0 * gL1 @ L1s, L2s - s0 * gL2 @ L1s - s0 * gL3 * L2s], 0)   # :463-464
        nl3 = l3s - s0 * gL3 * l3s                                                     # :465
        gU1 = np.triu(self.Pg1 @ g[:r].T - x[:r] @ iPx1.T)                             # :468-471
        gU2, gU3 = self.Pg1 @ g[r:].T - x[:r] @ self.iPx2.T, self.Pg2 * g[r:] - x[r:] * self.iPx2
        s0 = step / (max(np.max(np.abs(gU1)), m[1] if has_tail else 0.0) + tiny)       # :472-475
        nU = np.concatenate([U1s - U1s @ (s0 * gU1), U2s - U1s @ (s0 * gU2) - s0 * gU3.T * U2s], 1)   # :476-477
        nu3 = u3s - s0 * gU3 * u3s                                                     # :478
        return tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in (nL, nl3, nU, nu3))
