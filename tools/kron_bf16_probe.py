import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from tools.kron_timing import state
M = N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
from psgd_tf_amd import _lib
_lib.load().psgd_kron_bf16_set_tuning(0, int(os.environ.get("HG_VARIANT", "0")))
if os.environ.get("HG_PATCH"):          # tile rows of an XCD's patch in the fused pair (bf16 tuning key 7; 0 = whole tile columns)
    _lib.load().psgd_kron_bf16_set_tuning(7, int(os.environ["HG_PATCH"]))
Ql, Qr, dX, dG, G = state(M, N, torch.device("cuda:0"))
Gb = G.to(torch.bfloat16)
for _ in range(4):
    psgd.precond_grad_kron(Ql, Qr, Gb)
torch.cuda.synchronize()
