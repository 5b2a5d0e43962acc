"""A/B of the two-chain overlap of large Kron updates (kron tuning key 9): products of psgd.py:173 on a side stream next
to the solves of :174.  fp32 and bf16-operand updates over a grid of shapes (run on the GPU box)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import _lib
from tools.kron_timing import state

lib = _lib.load()
dev = torch.device("cuda:0")
SH = [(600, 530), (1000, 1000), (1024, 1024), (1536, 1536), (2048, 2048), (3072, 3072), (4096, 4096), (300, 4000),
      (4000, 300), (64, 8192), (2048, 4096), (1024, 4096)]
if len(sys.argv) > 1:
    SH = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]


def t(fn, min_ms=60.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < 40.0:      # steady clocks
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while True:
        fn(); n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
            if (time.perf_counter() - t0) * 1e3 > min_ms:
                break
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    for m, n in SH:
        Ql, Qr, dX, dG, G = state(m, n, dev)
        row = []
        for kind in ("f32", "bf16"):
            if kind == "bf16" and (m % 8 or n % 8):
                continue
            x, g = (dX, dG) if kind == "f32" else (dX.to(torch.bfloat16), dG.to(torch.bfloat16))
            res, tm = {}, {}
            for key in (0, 1, 0, 1):
                lib.psgd_kron_set_tuning(9, key)
                tm.setdefault(key, []).append(t(lambda: psgd.update_precond_kron(Ql, Qr, x, g, 0.01)))
                res[key] = [r.clone() for r in psgd.update_precond_kron(Ql, Qr, x, g, 0.01)]
            same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
            row.append(f"{kind}: serial {min(tm[0]):.3f} ms, overlapped {min(tm[1]):.3f} ms ({min(tm[0]) / min(tm[1]):.2f}x, equal {same})")
        print(f"{m} x {n}: " + " | ".join(row), flush=True)
    lib.psgd_kron_set_tuning(9, 1)


if __name__ == "__main__":
    main()
