"""Round 6, review item 9 (one bounded experiment): the wide Gram (uvd_wide_gram.hip, ranks 33..64) with the multiplying waves' fp64 sums
in LDS instead of registers (build flag GW_ACC64_LDS=1).  One process per build (PSGD_HIP_LIB), alternating; prints the time of
psgd_uvd_gram_wide_f32 at N rows and a checksum of G.     python tools/r06_wide_gram_ab.py <tag> [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgd_tf_amd import _lib  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "?"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for r in (40, 48, 64):
        g = torch.Generator(device=dev).manual_seed(r)
        sc = (1.0 / (N * r)) ** 0.5
        U, V = torch.randn(N, r, device=dev, generator=g) * sc, torch.randn(N, r, device=dev, generator=g) * sc
        d = torch.ones(N, 1, device=dev) * 1.3
        v = torch.randn(N, 1, device=dev, generator=g)
        h = v * 1.5
        nb = int(lib.psgd_uvd_gram_wide_scratch_bytes(N, r))
        scr = torch.empty(nb, dtype=torch.uint8, device=dev)
        G = torch.zeros(2 * r + 2, 2 * r + 2, dtype=torch.float64, device=dev)

        def call():
            rc = lib.psgd_uvd_gram_wide_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(), h.data_ptr(), N, r, G.data_ptr(),
                                            scr.data_ptr(), nb, st)
            assert rc == 0, rc
        for _ in range(3):
            call()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        out.append("r=%d %.3f ms (%.2f TB/s) sum %.17g" % (r, best, 8.0 * N * r / best / 1e9, float(G.sum())))
        del U, V
    print("[%s] N=%d  " % (tag, N) + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
