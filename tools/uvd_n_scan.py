"""UVd fused step / apply / update over N at fixed ranks (steady clocks): looks for sizes that fall off the curve (streaming-
policy and grid-size thresholds).   python tools/uvd_n_scan.py"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from bench import make_inputs, STEP, TINY  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for r in (10, 20):
        for N in (10_000, 30_000, 100_000, 300_000, 1_000_000, 2_000_000, 3_000_000, 4_000_000, 6_000_000, 8_000_000, 12_000_000,
                  16_000_000, 24_000_000, 32_000_000, 48_000_000, 64_000_000, 100_000_000):
            U, V, d, g, v, h = make_inputs(N, N, r, dev, 7)

            def timeit(fn, warm_ms=30.0, min_ms=30.0):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(2):
                    fn(i)
                e1.record()
                torch.cuda.synchronize()
                per = max(e0.elapsed_time(e1) / 2, 1e-3)
                for i in range(min(3000, int(warm_ms / per))):
                    fn(i)
                n = max(6, min(3000, int(min_ms / per) + 1))
                n += n % 2
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
            print("r = %2d N = %9d  fused %9.1f us %5.2f TB/s | apply %9.1f us %5.2f TB/s | update %9.1f us %5.2f TB/s" %
                  (r, N, tf * 1e3, bf * N / tf * 1e-9, ta * 1e3, ba * N / ta * 1e-9, tu * 1e3, bu * N / tu * 1e-9))
            del U, V, d, g, v, h
