"""Sparse-LU preconditioner update + apply (psgd.py:396-524) over every rank 1..32 at a fixed N: bytes the sweeps move over the
time (update 4 (9 r + 15), apply 4 (3 r + 9) bytes per parameter: DESIGN 4.5), steady clocks, to find ranks that fall off their neighbours.   python tools/splu_rank_scan.py [N]"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device("cuda:0")
    for r in range(1, 33):
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

        def timeit(fn, warm_ms=40.0, min_ms=40.0):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(2):
                fn()
            e1.record()
            torch.cuda.synchronize()
            per = max(e0.elapsed_time(e1) / 2, 1e-3)
            for i in range(int(warm_ms / per)):
                fn()
            n = max(6, int(min_ms / per) + 1)
            e0.record()
            for i in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        tu = timeit(lambda: psgd.update_precond_splu(L12, l3, U12, u3, [dx], [dg], 0.01))
        ta = timeit(lambda: psgd.precond_grad_splu(L12, l3, U12, u3, [gr]))
        bu, ba = 4 * (9 * r + 15), 4 * (3 * r + 9)
        print("r = %2d  update %7.3f ms %5.2f TB/s | apply %7.3f ms %5.2f TB/s" % (r, tu, bu * N / tu * 1e-9, ta, ba * N / ta * 1e-9))
        del L12, U12, l3, u3, dx, dg, gr
