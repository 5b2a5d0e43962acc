"""Round 6, review item 3: the products of psgd.py:173 on a background stream beside BOTH inversions (tuning key 30 = bytes of LDS padding that
keep one block of them per CU; 1 = the background stream without padding; 0 = behind Ql's inversion on the side stream, the round-5 order).
One process, the settings alternate; results compared with the setting 0's.   python tools/r06_kron_bg.py [M N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from r06_kron_ab import timeit  # noqa: E402


def main():
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(4096, 4096), (2048, 4096), (3072, 3072), (6144, 6144)]
    settings = [int(x) for x in os.environ.get("BG", "0,1,32768,0,1,32768").split(",")]
    key = int(os.environ.get("KEY", "30"))                  # (KEY=31: the balanced launches' smax, -1 = the plan's own choice)
    lib = _lib.load()
    dev = torch.device("cuda:0")
    for M, N in shapes:
        g = torch.Generator(device=dev).manual_seed(M * 7 + N)
        Ql = torch.triu(torch.randn(M, M, device=dev, generator=g) * 0.02, 1) + torch.eye(M, device=dev)
        Qr = torch.triu(torch.randn(N, N, device=dev, generator=g) * 0.02, 1) + torch.eye(N, device=dev)
        G, dX = torch.randn(M, N, device=dev, generator=g), torch.randn(M, N, device=dev, generator=g)
        if os.environ.get("APPLY"):                  # the apply with NEW factors on every call: default ("reference") and "auto" routes
            from psgd_tf_amd import kron
            Ql2, Qr2 = Ql.clone(), Qr.clone()
            pairs, flip = [(Ql, Qr), (Ql2, Qr2)], [0]

            def cold():
                flip[0] ^= 1
                return psgd.precond_grad_kron(pairs[flip[0]][0], pairs[flip[0]][1], G)
            line, ref_out = [], None
            for v in settings:
                assert lib.psgd_kron_set_tuning(key, v) == 0
                kron.invalidate_factor_cache()
                out = cold().clone()
                if ref_out is None:
                    ref_out = out
                tr = min(timeit(cold, 10) for _ in range(3))
                old = kron.set_apply_route("auto")
                ta = min(timeit(cold, 10) for _ in range(3))
                kron.set_apply_route(old)
                line.append("%d: ref %.3f auto %.3f ms (%.1e)" % (v, tr, ta, float((out - ref_out).abs().max() / ref_out.abs().max())))
            print("%dx%d apply  " % (M, N) + " | ".join(line), flush=True)
            continue
        ref = None
        for bf in (False, True):
            a, b = (dX.to(torch.bfloat16), G.to(torch.bfloat16)) if bf else (dX, G)
            line = []
            for v in settings:
                assert lib.psgd_kron_set_tuning(key, v) == 0
                out = psgd.update_precond_kron(Ql, Qr, a, b, 0.01)
                torch.cuda.synchronize()
                if ref is None:
                    ref = [o.clone() for o in out]
                err = max(float((o - r).abs().max()) for o, r in zip(out, ref)) if not bf else float("nan")
                t = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, a, b, 0.01), 5) for _ in range(3))
                line.append("%d: %.3f ms (%.1e)" % (v, t, err))
            print("%dx%d %s  " % (M, N, "bf16-ops" if bf else "fp32") + " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
