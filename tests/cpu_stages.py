"""CPU stage backend for tests/test_sharded_cpu.py.  TEST INFRASTRUCTURE ONLY.

Implements the stage interface of psgd_tf_amd.sharded.HipStages with NumPy, so that the
multi-rank choreography (which buffer is all-reduced when, SUM vs MAX, what each rank
recomputes redundantly) can run under gloo on a machine without a GPU.  The stage math restates
the fused plan of the HIP kernels: Gram of [U V t w] -> r x r algebra ->
row-local update; the test compares its sharded result with the reference-order oracle.
"""
import numpy as np
import torch


class NumpyStages:
    def __init__(self, r, dtype=np.float64):
        self.r = r
        self.dtype = dtype
        nc = 2 * r + 2
        self._sums = {1: torch.zeros(r, dtype=torch.float64), 2: torch.zeros(r, dtype=torch.float64),
                      11: torch.zeros(nc * nc, dtype=torch.float64), 13: torch.zeros(2 * r, dtype=torch.float64)}
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
        if update_U:
            c1, c2 = e1 @ K, e2 @ K
            Un -= mu * (a @ c(c1).T - b @ c(c2).T)
        else:
            al, be = a + Vn @ c(e1), b + Vn @ c(e2)
            Vn -= mu * (al @ c(e1).T - be @ c(e2).T)

    def update_sweep2_fused(self, U, V, d, v, h, g, step, tiny, update_U):
        """sweep 2 + the two extra column reductions of the fused update->apply."""
        dn, gn = self._np(d), self._np(g)
        d_old = dn.copy()
        self.update_sweep2(U, V, d, v, h, step, tiny, update_U)
        Vn = self._np(V)                                   # the NEW V in the V branch
        p = Vn.T @ (d_old * gn)
        q = Vn.T @ (d_old * gn * self.nabla)
        self._sums[13][:] = torch.from_numpy(np.concatenate([p, q]).astype(np.float64).ravel())

    def fused_s1(self, step, tiny):
        mu = step / (float(self._max[12][0]) + tiny)
        pq = self._sums[13].numpy()
        self._sums[1][:] = torch.from_numpy(pq[:self.r] - mu * pq[self.r:])

    def apply_sweep2_local_s1(self, U, d, g):
        self.apply_sweep2(U, d, g)

    def update_sweep3(self, d, step, tiny):
        dn = self._np(d)
        mu = step / (float(self._max[12][0]) + tiny)
        dn -= mu * dn * self.nabla
