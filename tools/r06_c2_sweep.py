"""Config 2 (N = 1M, r = 10): the fused step against the two launch-shape knobs of the sweeps (psgd_set_tuning key 3 = tiles a wave
should stream at least, key 1 = blocks per CU; negative = forced) -- VERDICT r5 item 6."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402

TINY = 1.1754943508222875e-38


def main():
    N, r = int(os.environ.get("C2_N", 1_000_000)), int(os.environ.get("C2_R", 10))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(7)
    sc = (1.0 / (N * r)) ** 0.5
    U, V = torch.randn(N, r, device=dev, generator=g) * sc, torch.randn(N, r, device=dev, generator=g) * sc
    d = torch.ones(N, 1, device=dev)
    gr, v = torch.randn(N, 1, device=dev, generator=g), torch.randn(N, 1, device=dev, generator=g)
    h = v * 1.5
    out = torch.empty_like(gr)

    def step(i):
        return psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, gr, 0.0, TINY, balance=False, update_U=(i % 2 == 0), out=out)

    def measure(iters=300):
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        lib.psgd_prof_enable(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            step(i)
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) / iters * 1e3
        lib.psgd_prof_enable(1)
        for i in range(100):
            step(i)
        torch.cuda.synchronize()
        ks = []
        for slot in (3, 4, 2):
            tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
            lib.psgd_prof_collect(slot, ctypes.byref(tot), ctypes.byref(cnt))
            ks.append(tot.value / max(cnt.value, 1) * 1e3)
        lib.psgd_prof_enable(0)
        return wall, ks
    for key3 in (8, 4, 2, 1):
        for key1 in (0, -1, -2, -3, -4):
            lib.psgd_set_tuning(3, key3)
            lib.psgd_set_tuning(1, key1)
            try:
                w, ks = measure()
            except Exception as exc:
                print("tiles/wave %d blocks/CU %d: %r" % (key3, key1, exc))
                continue
            print("tiles/wave>=%d blocks/CU %2d : step %6.1f us | gram %5.1f  s2 %5.1f  final %5.1f  (sum %5.1f)" %
                  (key3, key1, w, ks[0], ks[1], ks[2], sum(ks)), flush=True)
    lib.psgd_set_tuning(3, 8)
    lib.psgd_set_tuning(1, 0)


if __name__ == "__main__":
    main()
