"""Seeded synthetic UVd problems shared by the CPU and GPU tests (SURVEY 8d inputs)."""
import numpy as np

TINY32 = float(np.finfo(np.float32).tiny)


def make_uvd_problem(N, r, seed=0, uv_gain=1.0, d_spread=0.0):
    """U, V ~ N(0,1) * uv_gain * (N r)^-1/2 (psgd.py:687-689), d = exp(d_spread * N(0,1))
    (d = 1 at d_spread = 0, psgd.py:690), g, v ~ N(0,1) (psgd.py:713), h = c .* v with
    c ~ LogUniform[1e-2, 1e2] (a diagonal SPD Hessian).  All fp32 arrays; d,g,v,h are [N,1]."""
    rng = np.random.default_rng(seed)
    scale = uv_gain * (1.0 / (N * r)) ** 0.5
    U = (rng.standard_normal((N, r)) * scale).astype(np.float32)
    V = (rng.standard_normal((N, r)) * scale).astype(np.float32)
    d = np.exp(d_spread * rng.standard_normal((N, 1))).astype(np.float32)
    g = rng.standard_normal((N, 1)).astype(np.float32)
    v = rng.standard_normal((N, 1)).astype(np.float32)
    c = np.exp(rng.uniform(np.log(1e-2), np.log(1e2), size=(N, 1)))
    h = (c * v).astype(np.float32)
    return dict(U=U, V=V, d=d, g=g, v=v, h=h)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (den if den > 0 else 1.0))
