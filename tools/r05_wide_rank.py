"""Ranks 33 .. 64 against the specialised ranks: UVd and sparse-LU, apply and update, N rows, bytes the
reference's arithmetic needs per parameter over the time.   python tools/r05_wide_rank.py [N]   ->  profiles/r05_wide_rank.txt

UVd: apply 4 (4r + 5), update 4 (5r + 10) bytes per row; sparse-LU: apply 4 (3r + 9), update 4 (9r + 15).
Each configuration runs in its own process; the second block is the round-4 route (column chunks of width <= 32:
PSGD_WIDE_FULL=0, PSGD_SPLU_CHUNKS=1) on the same box.
"""
import os
import subprocess
import sys
import torch


def timeit(fn, warm=3, n=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def child(N):
    sys.path.insert(0, ".")
    import preconditioned_stochastic_gradient_descent as psgd
    dev = torch.device("cuda")
    tag = "chunks" if os.environ.get("PSGD_WIDE_FULL") == "0" else "whole"
    base = {}
    for r in (20, 32, 40, 48, 64):
        g = torch.Generator(device=dev).manual_seed(r)
        sc = (1.0 / (N * r)) ** 0.5
        U, V = torch.randn(N, r, device=dev, generator=g) * sc, torch.randn(N, r, device=dev, generator=g) * sc
        d = torch.ones(N, 1, device=dev)
        gr, v = torch.randn(N, 1, device=dev, generator=g), torch.randn(N, 1, device=dev, generator=g)
        h = v * 1.5
        ta = timeit(lambda: psgd.precond_grad_UVd_math(U, V, d, gr))
        tu = timeit(lambda: psgd.update_precond_UVd_math_(U, V, d, v, h, 0.01, 1e-38, balance=False, update_U=True))
        tv = timeit(lambda: psgd.update_precond_UVd_math_(U, V, d, v, h, 0.01, 1e-38, balance=False, update_U=False))
        ra, ru = 4 * (4 * r + 5) * N / ta / 1e9, 4 * (5 * r + 10) * N / min(tu, tv) / 1e9
        if r <= 32:
            base["uvd"] = (max(base.get("uvd", (0, 0))[0], ra), max(base.get("uvd", (0, 0))[1], ru))
        print("UVd   r=%2d  apply %6.2f ms %5.2f TB/s | update U %6.2f ms  V %6.2f ms  %5.2f TB/s%s   [%s]"
              % (r, ta, ra, tu, tv, ru, "" if r <= 32 else "   x%.2f / x%.2f of the best specialised rate" % (ra / base["uvd"][0], ru / base["uvd"][1]), tag))
        del U, V
    for r in (20, 32, 40, 48, 64):
        g = torch.Generator(device=dev).manual_seed(3)
        n2, sc = N - r, 0.3 / r ** 0.5
        L12 = torch.randn(N, r, device=dev, generator=g) * (sc * 3 * (r / N) ** 0.5)
        U12 = torch.randn(r, N, device=dev, generator=g) * (sc * 3 * (r / N) ** 0.5)
        L12[:r] = torch.tril(torch.randn(r, r, device=dev, generator=g) * sc, -1) + torch.eye(r, device=dev)
        U12[:, :r] = torch.triu(torch.randn(r, r, device=dev, generator=g) * sc, 1) + torch.eye(r, device=dev)
        l3 = torch.exp(torch.empty(n2, 1, device=dev).uniform_(-0.5, 0.5, generator=g))
        u3 = torch.exp(torch.empty(n2, 1, device=dev).uniform_(-0.5, 0.5, generator=g)) * 0.7
        dx = torch.randn(N, 1, device=dev, generator=g)
        dg = dx * torch.exp(torch.empty(N, 1, device=dev).uniform_(-2.3, 2.3, generator=g))
        gr = torch.randn(N, 1, device=dev, generator=g)
        tu = timeit(lambda: psgd.update_precond_splu(L12, l3, U12, u3, [dx], [dg], 0.01))
        ta = timeit(lambda: psgd.precond_grad_splu(L12, l3, U12, u3, [gr]))
        ra, ru = 4 * (3 * r + 9) * N / ta / 1e9, 4 * (9 * r + 15) * N / tu / 1e9
        if r <= 32:
            base["lu"] = (max(base.get("lu", (0, 0))[0], ra), max(base.get("lu", (0, 0))[1], ru))
        print("spLU  r=%2d  apply %6.2f ms %5.2f TB/s | update %6.2f ms %5.2f TB/s%s   [%s]"
              % (r, ta, ra, tu, ru, "" if r <= 32 else "   x%.2f / x%.2f of the best specialised rate" % (ra / base["lu"][0], ru / base["lu"][1]), tag))
        del L12, U12


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        child(int(sys.argv[1]))
        sys.exit(0)
    N = sys.argv[1] if len(sys.argv) > 1 else "20000000"
    print("N = %s rows" % N)
    for env in ({}, {"PSGD_WIDE_FULL": "0", "PSGD_SPLU_CHUNKS": "1"}):
        r = subprocess.run([sys.executable, __file__, N, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        print(r.stdout.strip() if r.returncode == 0 else (r.stdout + r.stderr[-3000:]))
