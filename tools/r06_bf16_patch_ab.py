"""bf16 Kron apply: the XCD-patch tile map of the fused triangular pair (psgd_kron_bf16_set_tuning key 7) against whole tile columns
per XCD -- time (new factors every call / unchanged factors) and bitwise equality of the outputs.  VERDICT r5 item 2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib, kron  # noqa: E402


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
    dev = torch.device("cuda:0")
    lib = _lib.load()
    shapes = [(4096, 4096), (2048, 4096), (4096, 2048), (2560, 2560), (2048, 2048), (1024, 2048), (6144, 6144), (8192, 2048)]
    for M, N in shapes:
        g = torch.Generator(device=dev).manual_seed(M * 7 + N)
        Ql = torch.triu(torch.randn(M, M, device=dev, generator=g) * 0.02, 1) + torch.eye(M, device=dev)
        Qr = torch.triu(torch.randn(N, N, device=dev, generator=g) * 0.02, 1) + torch.eye(N, device=dev)
        Gb = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        Ql2, Qr2 = Ql.clone(), Qr.clone()
        pairs, flip = [(Ql, Qr), (Ql2, Qr2)], [0]

        def cold():
            flip[0] ^= 1
            return psgd.precond_grad_kron(pairs[flip[0]][0], pairs[flip[0]][1], Gb)
        res, line = {}, []
        for rep in range(2):
            for patch in (0, 4, 8, 2):
                lib.psgd_kron_bf16_set_tuning(7, patch)
                out = psgd.precond_grad_kron(Ql, Qr, Gb).clone()
                res.setdefault(patch, out)
                assert torch.equal(res[patch], out)
                tc = min(timeit(cold, 20) for _ in range(2))
                tw = min(timeit(lambda: psgd.precond_grad_kron(Ql, Qr, Gb), 20) for _ in range(2))
                line.append("patch %d: %.4f / %.4f" % (patch, tc, tw))
        same = all(torch.equal(res[0], res[k]) for k in res)
        print("%5d x %-5d  new factors / unchanged (ms):  %s   bitwise equal: %s" % (M, N, "   ".join(line), same), flush=True)
    lib.psgd_kron_bf16_set_tuning(7, 4)


if __name__ == "__main__":
    main()
