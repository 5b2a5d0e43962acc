"""UVd fused step (update + apply) and the two calls alone over every rank 1..32 (+ a few wide ones) at a fixed N: bytes the
sweeps move over the time, to find ranks that fall off their neighbours.   python tools/uvd_rank_scan.py [N]"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from bench import make_inputs, STEP, TINY  # noqa: E402

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    dev = torch.device("cuda:0")
    ranks = list(range(1, 33)) + [40, 48, 64]
    if len(sys.argv) > 2:
        ranks = [int(x) for x in sys.argv[2].split(",")]
    print("N = %d; bytes per row: fused step 4 (2r+3) + 4 (3r+5) + 4 (2r+5), apply 4 (3r+8), update 4 (5r+7) + 12" % N)
    for r in ranks:
        if N * r * 4 * 2 > 60e9:
            continue
        U, V, d, g, v, h = make_inputs(N, N, r, dev, 7)

        def timeit(fn, n=6, warm_ms=40.0, min_ms=40.0):           # steady clocks: >= 40 ms of the call before, >= 40 ms timed
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(2):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            per = max(e0.elapsed_time(e1) / 2, 1e-3)
            for i in range(int(warm_ms / per)):
                fn(i)
            n = max(n, int(min_ms / per) + 1)
            n += n % 2                                              # both update branches equally often
            e0.record()
            for i in range(n):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        tf = timeit(lambda i: psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, g, STEP, TINY, balance=False, update_U=(i % 2 == 0)))
        ta = timeit(lambda i: psgd.precond_grad_UVd_math(U, V, d, g))
        tu = timeit(lambda i: psgd.update_precond_UVd_math_(U, V, d, v, h, STEP, TINY, balance=False, update_U=(i % 2 == 0)))
        bf = 4 * (2 * r + 3) + 4 * (3 * r + 5) + 4 * (2 * r + 5)
        ba, bu = 4 * (3 * r + 8), 4 * (5 * r + 7) + 12
        print("r = %2d  fused %7.3f ms %5.2f TB/s | apply %7.3f ms %5.2f TB/s | update %7.3f ms %5.2f TB/s" %
              (r, tf, bf * N / tf * 1e-9, ta, ba * N / ta * 1e-9, tu, bu * N / tu * 1e-9))
        del U, V, d, g, v, h
