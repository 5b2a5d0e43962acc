"""Round 6: do the large Kron calls care where their buffers live?  (The UVd sweeps do: DESIGN 4.1a.)  The same calls with B GiB of
ballast allocated first, so that factors, workspace and operands land B GiB further down the device's address map.
    python tools/r06_kron_region_probe.py [ballast GiB ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from r06_kron_ab import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M = N = 4096
    for gib in [float(x) for x in sys.argv[1:]] or [0.0, 40.0, 90.0, 130.0, 190.0]:
        torch.cuda.empty_cache()
        ballast = torch.empty(int(gib * 2**30), dtype=torch.uint8, device=dev) if gib > 0 else None
        from psgd_tf_amd import kron
        for c in (kron._kron_ws, kron._kron_ws_bf16):
            c._d.clear()
        g = torch.Generator(device=dev).manual_seed(7)
        Ql = torch.triu(torch.randn(M, M, device=dev, generator=g) * 0.02, 1) + torch.eye(M, device=dev)
        Qr = torch.triu(torch.randn(N, N, device=dev, generator=g) * 0.02, 1) + torch.eye(N, device=dev)
        G, dX = torch.randn(M, N, device=dev, generator=g), torch.randn(M, N, device=dev, generator=g)
        Gb = G.to(torch.bfloat16)
        tu = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, G, 0.01), 5) for _ in range(3))
        ta = min(timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), 10) for _ in range(3))
        tb = min(timeit(lambda: psgd.precond_grad_kron(Ql, Qr, Gb), 20) for _ in range(3))
        print("ballast %5.0f GiB (first tensor at %#x): fp32 update %.3f ms | fp32 apply (unchanged factors) %.3f | bf16 apply (unchanged) %.4f"
              % (gib, Ql.data_ptr(), tu, ta, tb), flush=True)
        del ballast, Ql, Qr, G, dX, Gb


if __name__ == "__main__":
    main()
