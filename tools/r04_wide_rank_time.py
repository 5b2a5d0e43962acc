"""Wide-rank (r > 32) UVd and sparse-LU paths: column VIEWS (round 4, the *_ld entry points) against the round-3 chunk copies
(PSGD_WIDE_COPIES=1), next to the specialised r = 20 path at the same N.   python tools/r04_wide_rank_time.py [N]"""
import os
import subprocess
import sys
import torch

if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, ".")
    import preconditioned_stochastic_gradient_descent as psgd
    N = int(sys.argv[1])
    dev = torch.device("cuda")

    def timeit(fn, n=6):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    for r in (20, 40, 64, 100):
        g = torch.Generator(device=dev).manual_seed(r)
        sc = (1.0 / (N * r)) ** 0.5
        U, V = torch.randn(N, r, device=dev, generator=g) * sc, torch.randn(N, r, device=dev, generator=g) * sc
        d = torch.ones(N, 1, device=dev)
        gr, v = torch.randn(N, 1, device=dev, generator=g), torch.randn(N, 1, device=dev, generator=g)
        h = v * 1.5
        ta = timeit(lambda: psgd.precond_grad_UVd_math(U, V, d, gr))
        tu = timeit(lambda: psgd.update_precond_UVd_math_(U, V, d, v, h, 0.01, 1e-38, balance=False, update_U=True))
        print("UVd  N=%d r=%3d  apply %.2f ms (%.2f TB/s on 4(4r+5) B/param)   update %.2f ms (%.2f TB/s on 4(5r+10))   [%s]"
              % (N, r, ta, 4 * (4 * r + 5) * N / ta / 1e9, tu, 4 * (5 * r + 10) * N / tu / 1e9,
                 "copies" if os.environ.get("PSGD_WIDE_COPIES") == "1" else "views"))
        del U, V
    sys.exit(0)

N = sys.argv[1] if len(sys.argv) > 1 else "20000000"
for env in ({}, {"PSGD_WIDE_COPIES": "1"}):
    r = subprocess.run([sys.executable, __file__, N, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-2000:])
