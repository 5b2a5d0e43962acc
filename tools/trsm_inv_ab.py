"""A/B of the solves of psgd.py:174: explicit inverses of the 2048-column diagonal groups (tuning key 11 = 1) against the
512-column substitution strips (0), fp32 and bf16-operand updates, interleaved rounds in one process.
   python tools/trsm_inv_ab.py [sizes ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import _lib
from tools.kron_timing import state

lib = _lib.load()
dev = torch.device("cuda:0")
sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(1024, 1024), (1536, 1536), (2048, 2048), (3072, 3072),
                                                                        (4096, 4096), (1100, 530), (4096, 1024), (1024, 4096)]


def t_of(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N in sizes:
    Ql, Qr, dX, dG, G = state(M, N, dev)
    dXb, dGb = dX.bfloat16(), dG.bfloat16()
    n = 40 if max(M, N) <= 2048 else 10
    res = {(i, b): [] for i in (1, 0) for b in (0, 1)}
    for rnd in range(3):
        for inv in (1, 0):
            lib.psgd_kron_set_tuning(11, inv)
            res[(inv, 0)].append(t_of(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), n))
            res[(inv, 1)].append(t_of(lambda: psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01), n))
    lib.psgd_kron_set_tuning(11, 1)
    med = lambda v: sorted(v)[len(v) // 2]
    print("%5d x %-5d fp32 update: inverses %.3f ms  strips %.3f ms  (%.2fx) | bf16 operands: inverses %.3f ms  strips %.3f ms  (%.2fx)"
          % (M, N, med(res[(1, 0)]), med(res[(0, 0)]), med(res[(0, 0)]) / med(res[(1, 0)]),
             med(res[(1, 1)]), med(res[(0, 1)]), med(res[(0, 1)]) / med(res[(1, 1)])), flush=True)
