"""Time one shape's Kron update under tuning keys from the environment: KRON_KEYS=25:1,10:0 python tools/r04_time_update.py 4096 4096 [bf16]"""
import os
import sys
import torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402
lib = _lib.load()
for kv in os.environ.get("KRON_KEYS", "").split(","):
    if ":" in kv:
        lib.psgd_kron_set_tuning(int(kv.split(":")[0]), int(kv.split(":")[1]))
M, N = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda"); g.manual_seed(1)
Ql, Qr = tri(M, g), tri(N, g)
dX = torch.randn(M, N, device="cuda", generator=g)
dG = dX * 1.5
if len(sys.argv) > 3 and sys.argv[3] == "bf16":
    dX, dG = dX.bfloat16(), dG.bfloat16()
t = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 8) for _ in range(3))
print("%dx%d %s KRON_KEYS=%s: %.3f ms" % (M, N, sys.argv[3] if len(sys.argv) > 3 else "f32", os.environ.get("KRON_KEYS", ""), t))
