import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from tools.kron_timing import state
M = N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Ql, Qr, dX, dG, G = state(M, N, torch.device("cuda:0"))
if len(sys.argv) > 2 and sys.argv[2] == "bf16":
    dX, dG = dX.bfloat16(), dG.bfloat16()
if os.environ.get("KRON_KEY"):                      # e.g. KRON_KEY=6:0 -> psgd_kron_set_tuning(6, 0)
    from psgd_tf_amd import _lib
    k, v = os.environ["KRON_KEY"].split(":")
    _lib.load().psgd_kron_set_tuning(int(k), int(v))
for _ in range(3):
    psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
torch.cuda.synchronize()
