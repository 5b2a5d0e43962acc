"""One shape's Kron update in a loop, for `rocprofv3 --kernel-trace --stats`:
    python tools/kron_update_trace.py M N key12 [reps] [bf16]
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri  # noqa: E402

if __name__ == "__main__":
    M, N, key = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    bf16 = len(sys.argv) > 5 and sys.argv[5] == "bf16"
    lib = _lib.load()
    lib.psgd_kron_set_tuning(12, key)
    import os
    for kv in os.environ.get("KRON_KEYS", "").split(","):          # e.g. KRON_KEYS=25:1,24:2048
        if ":" in kv:
            lib.psgd_kron_set_tuning(int(kv.split(":")[0]), int(kv.split(":")[1]))
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    Ql, Qr = tri(M, g), tri(N, g)
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = dX * 1.5
    if bf16:
        dX, dG = dX.bfloat16(), dG.bfloat16()
    for _ in range(reps):
        psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
    torch.cuda.synchronize()
