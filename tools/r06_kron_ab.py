"""A/B of two builds of the library on the large Kron calls (steady-state protocol of bench.py: >= 30 ms of the same call before a
timed region of >= 20 ms).  PSGD_HIP_LIB selects the build; run the script once per build, alternating.
  PSGD_HIP_LIB=build_ab/libpsgd_hip_old.so python tools/r06_kron_ab.py old ; python tools/r06_kron_ab.py new"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import kron  # noqa: E402


def timeit(fn, n, warm_ms=30.0, min_ms=20.0):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    torch.cuda.synchronize()
    per = max(e0.elapsed_time(e1) / 3, 1e-3)
    for _ in range(min(2000, int(warm_ms / per))):
        fn()
    n = max(n, min(2000, int(min_ms / per) + 1))
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "?"
    dev = torch.device("cuda:0")
    out = []
    for M, N in ((4096, 4096), (2048, 4096), (6144, 6144)):
        g = torch.Generator(device=dev).manual_seed(M * 7 + N)
        Ql = torch.triu(torch.randn(M, M, device=dev, generator=g) * 0.02, 1) + torch.eye(M, device=dev)
        Qr = torch.triu(torch.randn(N, N, device=dev, generator=g) * 0.02, 1) + torch.eye(N, device=dev)
        G, dX = torch.randn(M, N, device=dev, generator=g), torch.randn(M, N, device=dev, generator=g)
        tu = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, G, 0.01), 5) for _ in range(3))
        Ql2, Qr2 = Ql.clone(), Qr.clone()
        pairs, flip = [(Ql, Qr), (Ql2, Qr2)], [0]

        def cold():
            flip[0] ^= 1
            return psgd.precond_grad_kron(pairs[flip[0]][0], pairs[flip[0]][1], G)
        ta_ref = min(timeit(cold, 10) for _ in range(3))
        old = kron.set_apply_route("auto")
        ta_auto = min(timeit(cold, 10) for _ in range(3))
        kron.set_apply_route(old)
        Gb, dXb = G.to(torch.bfloat16), dX.to(torch.bfloat16)
        tub = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dXb, Gb, 0.01), 5) for _ in range(3))
        out.append("%dx%d upd %.3f  upd(bf16 ops) %.3f  apply(ref, new factors) %.3f  apply(auto, new factors) %.3f" %
                   (M, N, tu, tub, ta_ref, ta_auto))
    print("[%s] " % tag + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
