"""Forward error of the triangular solve y T = x (psgd.py:174) through EXPLICIT fp32 inverses of diagonal groups of T (recursive
doubling from exactly inverted 32 x 32 blocks: [A B; 0 C]^-1 = [A^-1, -A^-1 B C^-1; 0, C^-1]) against fp32 substitution, both
measured against the fp64 solution for the fp32-rounded T.  CPU / NumPy model of what csrc/psgd_kron.hip (tri_group_inverses,
trsm_ut with an InvCtx) does on the device; group size S = n is the full inverse.  Output: profiles/r03_group_inverse_error_study.txt
   python tools/group_inverse_error_study.py"""
import numpy as np
rng = np.random.default_rng(0)
def illcond(n, kappa):
    d = np.exp(np.linspace(0.0, -np.log(kappa), n)); rng.shuffle(d)
    return np.triu(rng.standard_normal((n, n)) * (0.3 / n ** 0.5), 1) * d[None, :] + np.diag(d)
def tri_inv_rec32(T):   # recursive doubling in fp32 from exact 32-block inverses (like the device would)
    n = T.shape[0]; Inv = np.zeros_like(T)
    for b0 in range(0, n, 32):
        Inv[b0:b0+32, b0:b0+32] = np.linalg.inv(T[b0:b0+32, b0:b0+32].astype(np.float64)).astype(np.float32)
    b = 32
    while b < n:
        for p in range(0, n, 2*b):
            A = Inv[p:p+b, p:p+b]; C = Inv[p+b:p+2*b, p+b:p+2*b]; B = T[p:p+b, p+b:p+2*b]
            Inv[p:p+b, p+b:p+2*b] = -(A @ (B @ C))
        b *= 2
    return Inv
def solve_blocked(T, X, S):  # y T = x, right-looking over S-wide groups with explicit group inverses (fp32)
    n = T.shape[0]; Y = X.copy()
    for g in range(0, n, S):
        Inv = tri_inv_rec32(T[g:g+S, g:g+S])
        Y[:, g:g+S] = Y[:, g:g+S] @ Inv
        if g + S < n:
            Y[:, g+S:] -= Y[:, g:g+S] @ T[g:g+S, g+S:]
    return Y
import scipy.linalg as sl
for kappa in (1e1, 1e2, 1e3, 1e4, 1e5):
    n, m = 1024, 256
    T64 = illcond(n, kappa); T = T64.astype(np.float32)
    X = rng.standard_normal((m, n)).astype(np.float32)
    ref = sl.solve_triangular(T.astype(np.float64), X.astype(np.float64).T, trans='T', lower=False).T
    sub = sl.solve_triangular(T, X.T, trans='T', lower=False).T  # fp32 substitution (LAPACK strsm)
    out = {S: solve_blocked(T, X, S) for S in (32, 128, 512, 1024)}
    e = lambda y: np.linalg.norm(y - ref) / np.linalg.norm(ref)
    print("kappa %.0e cond2 %.1e | fp32 substitution %.1e | group inverse S=32 %.1e S=128 %.1e S=512 %.1e S=1024 %.1e"
          % (kappa, np.linalg.cond(T64), e(sub), e(out[32]), e(out[128]), e(out[512]), e(out[1024])))
print("--- Cholesky factors of SPD matrices with random eigenvectors (what a converged PSGD factor looks like): T'T = H^-1")
for kH in (1e2, 1e4, 1e6, 1e8):
    n, m = 1024, 256
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.exp(np.linspace(0, np.log(kH), n))
    Hinv = (V / lam) @ V.T
    T64 = np.linalg.cholesky(Hinv).T          # upper, T'T = H^-1
    T = T64.astype(np.float32)
    X = rng.standard_normal((m, n)).astype(np.float32)
    ref = sl.solve_triangular(T.astype(np.float64), X.astype(np.float64).T, trans='T', lower=False).T
    sub = sl.solve_triangular(T, X.T, trans='T', lower=False).T
    out = {S: solve_blocked(T, X, S) for S in (32, 128, 512, 1024)}
    e = lambda y: np.linalg.norm(y - ref) / np.linalg.norm(ref)
    print("kappa(H) %.0e cond2(T) %.1e | fp32 substitution %.1e | group inverse S=32 %.1e S=128 %.1e S=512 %.1e S=1024 %.1e"
          % (kH, np.linalg.cond(T64), e(sub), e(out[32]), e(out[128]), e(out[512]), e(out[1024])))
print("--- unit diagonal, N(0,1) c/sqrt(n) above it (inverse entries grow with c)")
for c in (0.3, 1.0, 2.0, 3.0, 4.0):
    n, m = 1024, 256
    T64 = np.triu(rng.standard_normal((n, n)) * (c / n ** 0.5), 1) + np.eye(n)
    T = T64.astype(np.float32)
    X = rng.standard_normal((m, n)).astype(np.float32)
    ref = sl.solve_triangular(T.astype(np.float64), X.astype(np.float64).T, trans='T', lower=False).T
    sub = sl.solve_triangular(T, X.T, trans='T', lower=False).T
    out = {S: solve_blocked(T, X, S) for S in (32, 128, 512, 1024)}
    e = lambda y: np.linalg.norm(y - ref) / np.linalg.norm(ref)
    print("c %.1f cond2(T) %.1e | fp32 substitution %.1e | group inverse S=32 %.1e S=128 %.1e S=512 %.1e S=1024 %.1e"
          % (c, np.linalg.cond(T64), e(sub), e(out[32]), e(out[128]), e(out[512]), e(out[1024])))
print("--- full inverse (S = n) with and without one refinement step (r = x - y T; y += r Inv), harder cases")
def study(T64, tag):
    n = T64.shape[0]; m = 256
    T = T64.astype(np.float32)
    X = rng.standard_normal((m, n)).astype(np.float32)
    ref = sl.solve_triangular(T.astype(np.float64), X.astype(np.float64).T, trans='T', lower=False).T
    sub = sl.solve_triangular(T, X.T, trans='T', lower=False).T
    Inv = tri_inv_rec32(T)
    y1 = X @ Inv
    r = X - y1 @ T
    y2 = y1 + r @ Inv
    e = lambda y: np.linalg.norm(y - ref) / np.linalg.norm(ref)
    k1 = np.abs(T).sum(0).max() * np.abs(Inv).sum(0).max()
    print("%s cond2 %.1e kappa1(est from fp32 Inv) %.1e | substitution %.1e | inverse %.1e | + 1 refinement %.1e" % (tag, np.linalg.cond(T64), k1, e(sub), e(y1), e(y2)))
n = 1024
for kH in (1e8, 1e10, 1e12, 1e14):
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.exp(np.linspace(0, np.log(kH), n))
    study(np.linalg.cholesky((V / lam) @ V.T).T, "chol kappa(H) %.0e" % kH)
for c in (4.0, 5.0, 6.0):
    study(np.triu(rng.standard_normal((n, n)) * (c / n ** 0.5), 1) + np.eye(n), "unit-diag c %.1f" % c)
# Kahan-type: T = diag(1, s, s^2..) * (I - c * strict_upper_ones)
for theta in (1.2, 1.0, 0.8):
    c, s = np.cos(theta), np.sin(theta); nn = 256
    K = np.diag(s ** np.arange(nn)) @ (np.eye(nn) - c * np.triu(np.ones((nn, nn)), 1))
    study(np.pad(K, ((0, n - nn), (0, n - nn))) + np.diag(np.r_[np.zeros(nn), np.ones(n - nn)]), "Kahan(256) theta %.1f" % theta)
