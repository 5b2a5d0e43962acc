"""A/B of where a phase of the 256^2 bf16 kernels issues its LDS-DMA pieces (psgd_kron_bf16_set_tuning key 4; hg256_mainloop):
0 = load half of the phase (the published template), 1 = both pieces inside the MFMA cluster, 2 = one and one.
Interleaved rounds in ONE process (results must be bit-identical: same MFMAs in the same order per accumulator).
   python tools/bf16_dmapos_ab.py [M N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
g = torch.Generator(device=dev).manual_seed(3)
tri = lambda n: torch.triu(torch.randn(n, n, device=dev, generator=g) * 0.02, 1) + torch.eye(n, device=dev)
Ql, Qr = tri(M), tri(N)
G = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)


def t_of(fn, n=40):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fn = lambda: psgd.precond_grad_kron(Ql, Qr, G)
ref = {}
for chain, name in ((1, "two fused pairs"), (0, "Gram-first chain (dense 256^2 kernel + one pair)")):
    lib.psgd_kron_bf16_set_tuning(1, chain)
    res = {0: [], 1: [], 2: []}
    for rnd in range(5):
        for pos in (0, 1, 2):
            lib.psgd_kron_bf16_set_tuning(4, pos)
            out = fn()
            if (chain, 0) not in ref and pos == 0:
                ref[(chain, 0)] = out.clone()
            assert torch.equal(out, ref[(chain, 0)]), "variant %d changed the result" % pos
            res[pos].append(t_of(fn))
    lib.psgd_kron_bf16_set_tuning(4, 0)
    print("%dx%d bf16 apply, %s:" % (M, N, name))
    for pos in (0, 1, 2):
        r = sorted(res[pos])
        print("   DMA placement %d: median %.1f us  min %.1f us  (rounds %s)" % (pos, r[len(r) // 2], r[0], " ".join("%.1f" % x for x in res[pos])))
lib.psgd_kron_bf16_set_tuning(1, 1)
