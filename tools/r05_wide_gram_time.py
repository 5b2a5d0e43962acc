"""Check and time psgd_uvd_gram_wide_f32 (the one-sweep Gram of ranks 33 .. 64, csrc/uvd_wide_gram.hip).

  python tools/r05_wide_gram_time.py [N] [ranks ...]

For each rank: (1) the Gram of a ragged N (1 000 003 rows) against the fp64 product of the same matrices (printed as max |dG| / scale,
the bound is that of a bf16 x 3 split: a few 1e-7 of sqrt(G_ii G_jj)), (2) the launch time at N rows (default 20 M) and the U + V
bytes it streams per second.  Output is the table of profiles/r05_wide_rank.txt.
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgd_tf_amd import _lib   # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")


def gram(U, V, d, v, h):
    N, r = U.shape
    n = int(lib.psgd_uvd_gram_wide_scratch_bytes(N, r))
    assert n > 0, n
    scr = torch.empty(n, dtype=torch.uint8, device=dev)
    G = torch.empty(2 * r + 2, 2 * r + 2, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def f():
        rc = lib.psgd_uvd_gram_wide_f32(U.data_ptr(), V.data_ptr(), d.data_ptr(), v.data_ptr(), h.data_ptr(), N, r, G.data_ptr(),
                                        scr.data_ptr(), n, st)
        assert rc == 0, rc
    return f, G


def check(r, N=1_000_003):
    g = torch.Generator(device=dev).manual_seed(r)
    U = torch.randn(N, r, device=dev, generator=g) * 3e-3
    V = torch.randn(N, r, device=dev, generator=g) * 2e-2
    d = torch.rand(N, device=dev, generator=g) + 0.5
    v = torch.randn(N, device=dev, generator=g)
    h = torch.randn(N, device=dev, generator=g)
    f, G = gram(U, V, d, v, h)
    f()
    torch.cuda.synchronize()
    W = torch.cat([U.double(), V.double(), (d * h).double()[:, None], (v / d).double()[:, None]], 1)
    ref = W.T @ W
    sc = ref.diagonal().sqrt()
    return float(((G - ref).abs() / (sc[:, None] * sc[None, :])).max())


def timeit(r, N, reps=5):
    U = torch.randn(N, r, device=dev) * 1e-4
    V = torch.randn(N, r, device=dev) * 1e-4
    d = torch.ones(N, device=dev)
    v = torch.randn(N, device=dev)
    h = v * 1.5
    f, _ = gram(U, V, d, v, h)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    ranks = [int(a) for a in sys.argv[2:]] or [33, 40, 48, 56, 64]
    print("psgd_uvd_gram_wide_f32, N = %d rows (accuracy on 1 000 003 rows)" % N)
    print("%4s %12s %10s %8s" % ("r", "max rel dG", "ms", "TB/s"))
    for r in ranks:
        err = check(r)
        ms = timeit(r, N)
        print("%4d %12.2e %10.3f %8.2f" % (r, err, ms, 2 * N * r * 4 / ms / 1e9))


if __name__ == "__main__":
    main()
