"""Round 6: does a LARGER tile per wave help the in-place read/write sweeps?  The rank-2 update kernel (M <- M - (a c1' - b c2'): reads M and
two vectors, rewrites M in place) on ONE 8-GB buffer seen as [100M, 20] (5-KiB tiles), [50M, 40] (10 KiB) and [31.25M, 64] (16 KiB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    buf = torch.zeros(2_000_000_000, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def timeit(fn, n=6):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    for r in (20, 40, 64, 32, 16):
        N = buf.numel() // r
        a, b = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        c = torch.zeros(2 * r, device=dev)
        M = buf[:N * r]
        if r <= 32:
            ws = psgd.uvd_workspace(dev, N, r)
            for grid in (256, 512, 768, 1024, 2048):
                lib.psgd_set_tuning(10 + 3, grid)
                t = timeit(lambda: _lib.check(lib.psgd_uvd_rank2_update_f32(M.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(), N, r,
                                                                            ws.data_ptr(), ws.numel(), st), "rank2"))
                print("r %2d (tile %5.1f KiB)  %4d workgroups : %.3f ms  %.0f GB/s" % (r, 64 * r * 4 / 1024 * (2 if r == 10 else 1), grid, t,
                                                                                   (8 * r + 8) * N / t / 1e6), flush=True)
            lib.psgd_set_tuning(10 + 3, 0)
        else:
            t = timeit(lambda: _lib.check(lib.psgd_uvd_wide_rank2_update_f32(M.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(), N, r, st),
                                          "wide rank2"))
            print("r %2d (tile %5.1f KiB)   512 workgroups : %.3f ms  %.0f GB/s" % (r, 64 * r * 4 / 1024, t, (8 * r + 8) * N / t / 1e6), flush=True)
        del a, b, c


if __name__ == "__main__":
    main()
