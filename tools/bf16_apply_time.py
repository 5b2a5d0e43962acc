"""Time of the bf16-operand Kron apply (what-if libraries: PSGD_HIP_LIB=tools/micro/libpsgd_hgN.so).   python tools/bf16_apply_time.py [n]"""
import os
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import kron  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    Ql, Qr = tri(n, g), tri(n, g)
    G = torch.randn(n, n, device="cuda", generator=g).bfloat16()
    t = min(timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), 30) for _ in range(3))
    print("%s  bf16 apply %d^2 (prepared factors): %.3f ms" % (os.environ.get("PSGD_HIP_LIB", "library"), n, t))
