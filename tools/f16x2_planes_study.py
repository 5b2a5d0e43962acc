"""CPU emulation that priced the accuracy of the "f16 x 2" plane format (csrc/psgd_kron.hip, PlaneMeta) before it was built:
the apply chain  (Ql'Ql) G (Qr'Qr)  with every product done as  h h' + h m' + m h'  on fp16 planes (fp64 accumulation
stands in for the matrix cores' fp32), against the bf16 x 3 split, plain fp32 and an fp64 reference.  Variants: scales
from compounded a-priori bounds, from exact maxima, per row / column, from K max|A| max|B| with the actual maxima of the
two inputs (c), and the same with the residual plane stored as M = 2^11 m (h) -- the one that was built.

    python tools/f16x2_planes_study.py  >  profiles/r03_f16x2_planes_study.txt
"""
import numpy as np


class orc:                                   # fp64 reference of the apply (psgd.py:388-391, dense x dense)
    @staticmethod
    def precond_grad_kron(Ql, Qr, G):
        return (Ql.T @ Ql) @ G @ (Qr.T @ Qr)


rng=np.random.default_rng(0)
def e_of(mx, K=1):
    # exponent e with max|x| * 2^e <= 2^14, and room for a K-term bound
    return 14 - int(np.ceil(np.log2(max(mx,1e-300)))) - int(np.ceil(np.log2(K)))
def split2(x, e):
    xs = (x.astype(np.float32) * np.float32(2.0**e)).astype(np.float32)
    h = xs.astype(np.float16)
    m = (xs - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float64), m.astype(np.float64)
def split3(x):
    x=x.astype(np.float32)
    def top(v): return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    h=top(x.copy()); r=x-h; m=top(r.copy()); l=r-m
    return h.astype(np.float64), m.astype(np.float64), top(l.copy()).astype(np.float64)
class P:  # planes with scale exponent and a bound on max|x|
    def __init__(s, x, bound=None):
        s.bound = float(np.max(np.abs(x))) if bound is None else bound
        s.e = e_of(s.bound); s.h, s.m = split2(x, s.e)
def mm2(A, B, trunc=True):   # A: P (M,K), B: P (K,N) given as arrays in natural orientation
    acc = A.h@B.h + A.h@B.m + A.m@B.h
    return (acc / 2.0**(A.e+B.e)).astype(np.float32)   # fp32 result (rounded)
def mm3(a, b):
    ah,am,al=split3(a); bh,bm,bl=split3(b)
    return (ah@bh+ah@bm+am@bh+ah@bl+al@bh+am@bm).astype(np.float32)
def rel(a,b): return np.linalg.norm(a-b)/np.linalg.norm(b)
def tri(n, off=0.05, spread=0.3): return np.triu(rng.standard_normal((n,n))*off,1)+np.diag(np.exp(spread*rng.standard_normal(n)))
def apply_chain(Ql,Qr,G,mode):
    f=lambda x: x.astype(np.float32)
    if mode=='f16':
        # association of planes_apply (M >= N): Pr = Qr'Qr ; T = G Pr ; U = Ql T ; out = Ql' U ; chained intermediates get bound-based scales
        pQr=P(Qr); K=Qr.shape[0]
        Pr=mm2(P(Qr.T),pQr); pPr=P(Pr, bound=K*pQr.bound**2)
        pG=P(G); T=mm2(pG,pPr); pT=P(T, bound=G.shape[1]*pG.bound*pPr.bound)
        pQl=P(Ql); U=mm2(pQl,pT); pU=P(U, bound=Ql.shape[0]*pQl.bound*pT.bound)
        return mm2(P(Ql.T),pU)
    if mode=='bf16x3':
        Pr=mm3(Qr.T,Qr); T=mm3(G,Pr); U=mm3(Ql,T); return mm3(Ql.T,U)
    if mode=='f32':
        Pr=f(Qr.T)@f(Qr); T=f(G)@Pr; U=f(Ql)@T; return f(Ql.T)@U
for (M,N,desc,mk) in [(512,384,'well-conditioned',lambda n:tri(n)), (512,384,'diag spread 1e4',lambda n:tri(n)*np.exp(np.linspace(0,-np.log(1e4),n))[None,:]),
                      (512,384,'strong off-diagonal',lambda n:tri(n,0.5)), (768,768,'cholesky cond 1e3',None)]:
    if mk is None:
        def mk(n):
            V,_=np.linalg.qr(rng.standard_normal((n,n))); lam=np.exp(np.linspace(0,np.log(1e6),n))
            Q=np.linalg.cholesky((V/lam)@V.T).T; return Q/np.max(np.abs(Q))
    Ql,Qr=mk(M).astype(np.float32),mk(N).astype(np.float32)
    G=(rng.standard_normal((M,N))*np.exp(rng.uniform(-3,3,(M,1)))).astype(np.float32)   # rows of very different scale
    ref=orc.precond_grad_kron(Ql.astype(np.float64),Qr.astype(np.float64),G.astype(np.float64))
    print('%-22s %dx%d | fp32 %.2e | bf16x3 %.2e | fp16 x2 (bound scales) %.2e' % (desc,M,N, rel(apply_chain(Ql,Qr,G,'f32'),ref), rel(apply_chain(Ql,Qr,G,'bf16x3'),ref), rel(apply_chain(Ql,Qr,G,'f16'),ref)))
print('--- exact per-matrix max scales (every intermediate re-split from its fp32 values with its own max)')
def apply_exact(Ql,Qr,G):
    Pr=mm2(P(Qr.T),P(Qr)); T=mm2(P(G),P(Pr)); U=mm2(P(Ql),P(T)); return mm2(P(Ql.T),P(U))
class PR:  # per-row (A operand) / per-column (B operand) scales
    def __init__(s, x, axis):
        mx=np.max(np.abs(x),axis=axis,keepdims=True); mx[mx==0]=1.0
        s.e = 14 - np.ceil(np.log2(mx)); xs=(x.astype(np.float32)*(2.0**s.e).astype(np.float32)).astype(np.float32)
        s.h=xs.astype(np.float16); s.m=(xs-s.h.astype(np.float32)).astype(np.float16); s.h=s.h.astype(np.float64); s.m=s.m.astype(np.float64)
def mmr(A,B):   # A: (M,K) row-scaled, B: (K,N) column-scaled
    pa,pb=PR(A,1),PR(B,0)
    acc=pa.h@pb.h+pa.h@pb.m+pa.m@pb.h
    return (acc/2.0**pa.e/2.0**pb.e).astype(np.float32)
def apply_rowcol(Ql,Qr,G):
    Pr=mmr(Qr.T,Qr); T=mmr(G,Pr); U=mmr(Ql,T); return mmr(Ql.T,U)
rng=np.random.default_rng(0)
for (M,N,desc,mk) in [(512,384,'well-conditioned',lambda n:tri(n)), (512,384,'diag spread 1e4',lambda n:tri(n)*np.exp(np.linspace(0,-np.log(1e4),n))[None,:]),
                      (512,384,'strong off-diagonal',lambda n:tri(n,0.5)), (768,768,'cholesky cond 1e3',None)]:
    if mk is None:
        def mk(n):
            V,_=np.linalg.qr(rng.standard_normal((n,n))); lam=np.exp(np.linspace(0,np.log(1e6),n))
            Q=np.linalg.cholesky((V/lam)@V.T).T; return Q/np.max(np.abs(Q))
    Ql,Qr=mk(M).astype(np.float32),mk(N).astype(np.float32)
    G=(rng.standard_normal((M,N))*np.exp(rng.uniform(-3,3,(M,1)))).astype(np.float32)
    ref=orc.precond_grad_kron(Ql.astype(np.float64),Qr.astype(np.float64),G.astype(np.float64))
    print('%-22s %dx%d | bf16x3 %.2e | fp16x2 exact matrix max %.2e | fp16x2 per-row/col scales %.2e' % (desc,M,N, rel(apply_chain(Ql,Qr,G,'bf16x3'),ref), rel(apply_exact(Ql,Qr,G),ref), rel(apply_rowcol(Ql,Qr,G),ref)))
print('--- (c): scale of an intermediate from the bound K * max|A| * max|B| with the ACTUAL maxima of its two inputs (no compounding)')
def apply_c(Ql,Qr,G):
    mx=lambda x: float(np.max(np.abs(x)))
    Pr=mm2(P(Qr.T),P(Qr)); pPr=P(Pr,bound=Qr.shape[0]*mx(Qr)**2)
    T=mm2(P(G),pPr); pT=P(T,bound=G.shape[1]*mx(G)*mx(Pr))
    U=mm2(P(Ql),pT); pU=P(U,bound=Ql.shape[0]*mx(Ql)*mx(T))
    return mm2(P(Ql.T),pU)
rng=np.random.default_rng(0)
for (M,N,desc,mk) in [(512,384,'well-conditioned',lambda n:tri(n)), (512,384,'diag spread 1e4',lambda n:tri(n)*np.exp(np.linspace(0,-np.log(1e4),n))[None,:]),
                      (512,384,'strong off-diagonal',lambda n:tri(n,0.5)), (768,768,'cholesky cond 1e3',None), (2048,2048,'well-conditioned',lambda n:tri(n,0.02))]:
    if mk is None:
        def mk(n):
            V,_=np.linalg.qr(rng.standard_normal((n,n))); lam=np.exp(np.linspace(0,np.log(1e6),n))
            Q=np.linalg.cholesky((V/lam)@V.T).T; return Q/np.max(np.abs(Q))
    Ql,Qr=mk(M).astype(np.float32),mk(N).astype(np.float32)
    G=(rng.standard_normal((M,N))*np.exp(rng.uniform(-3,3,(M,1)))).astype(np.float32)
    ref=orc.precond_grad_kron(Ql.astype(np.float64),Qr.astype(np.float64),G.astype(np.float64))
    print('%-22s %dx%d | bf16x3 %.2e | fp16x2 (c) %.2e | exact max %.2e' % (desc,M,N, rel(apply_chain(Ql,Qr,G,'bf16x3'),ref), rel(apply_c(Ql,Qr,G),ref), rel(apply_exact(Ql,Qr,G),ref)))
print('--- (h): residual plane stored as M = 2^11 m (never subnormal while h is normal); products h h\' + (2^-11 h) M\' + M (2^-11 h\')')
class PH:
    def __init__(s, x, bound=None):
        s.bound = float(np.max(np.abs(x))) if bound is None else bound
        s.e = e_of(s.bound)
        xs = (x.astype(np.float32) * np.float32(2.0**s.e)).astype(np.float32)
        h = xs.astype(np.float16)
        M = ((xs - h.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        H2 = (h.astype(np.float32) * np.float32(2.0**-11)).astype(np.float16)       # on the fly, in fp16
        s.h, s.M, s.H2 = h.astype(np.float64), M.astype(np.float64), H2.astype(np.float64)
def mmh(A,B):
    return ((A.h@B.h + A.H2@B.M + A.M@B.H2) / 2.0**(A.e+B.e)).astype(np.float32)
def apply_h(Ql,Qr,G,track):
    mx=lambda x: float(np.max(np.abs(x)))
    pQr,pQrT,pG,pQl,pQlT=PH(Qr),PH(Qr.T),PH(G),PH(Ql),PH(Ql.T)
    Pr=mmh(pQrT,pQr); bPr=Qr.shape[0]*pQr.bound**2; pPr=PH(Pr,bound=bPr)
    T=mmh(pG,pPr); bT=G.shape[1]*pG.bound*(mx(Pr) if track else bPr); pT=PH(T,bound=bT)
    U=mmh(pQl,pT); bU=Ql.shape[0]*pQl.bound*(mx(T) if track else bT); pU=PH(U,bound=bU)
    return mmh(pQlT,pU)
rng=np.random.default_rng(0)
for (M,N,desc,mk) in [(512,384,'diag spread 1e4',lambda n:tri(n)*np.exp(np.linspace(0,-np.log(1e4),n))[None,:]),
                      (768,768,'cholesky cond 1e3',None), (2048,2048,'well-conditioned',lambda n:tri(n,0.02))]:
    if mk is None:
        def mk(n):
            V,_=np.linalg.qr(rng.standard_normal((n,n))); lam=np.exp(np.linspace(0,np.log(1e6),n))
            Q=np.linalg.cholesky((V/lam)@V.T).T; return Q/np.max(np.abs(Q))
    Ql,Qr=mk(M).astype(np.float32),mk(N).astype(np.float32)
    G=(rng.standard_normal((M,N))*np.exp(rng.uniform(-3,3,(M,1)))).astype(np.float32)
    ref=orc.precond_grad_kron(Ql.astype(np.float64),Qr.astype(np.float64),G.astype(np.float64))
    print('%-22s %dx%d | bf16x3 %.2e | (h) bounds from actual input maxima %.2e | (h) bounds compounded, no tracking %.2e' % (desc,M,N, rel(apply_chain(Ql,Qr,G,'bf16x3'),ref), rel(apply_h(Ql,Qr,G,True),ref), rel(apply_h(Ql,Qr,G,False),ref)))
