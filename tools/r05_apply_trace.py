"""One shape's fp32 apply with new factors in a loop, for rocprofv3 --kernel-trace:  python tools/r05_apply_trace.py M N key28 [reps]"""
import sys
import torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import kron  # noqa: E402
from tools.kron_bf16_update_timing import tri  # noqa: E402
M, N, key = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
kron.set_tuning(28, key)
g = torch.Generator(device="cuda"); g.manual_seed(0)
Ql, Qr = tri(M, g), tri(N, g)
G = torch.randn(M, N, device="cuda", generator=g)
pairs = [(Ql.clone(), Qr.clone()) for _ in range(2)]
for i in range(reps):
    psgd.precond_grad_kron(pairs[i & 1][0], pairs[i & 1][1], G)
    kron.invalidate_factor_cache()
torch.cuda.synchronize()
