"""Per-kernel trace of the small-N (config 2) UVd calls under rocprofv3 (development aid).
    cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -o c2 -- python3 <repo>/tools/c2_trace.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from bench import make_inputs, STEP, TINY  # noqa: E402

from psgd_tf_amd import _lib  # noqa: E402
if os.environ.get("C2_TPW"):
    _lib.load().psgd_set_tuning(3, int(os.environ["C2_TPW"]))
N, r = int(os.environ.get("C2_N", "1000000")), int(os.environ.get("C2_R", "10"))
dev = torch.device("cuda:0")
U, V, d, g, v, h = make_inputs(N, N, r, dev, 7)
for i in range(60):
    psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, g, STEP, TINY, balance=False, update_U=(i % 2 == 0))
torch.cuda.synchronize()
if os.environ.get("C2_UNFUSED"):
    for i in range(60):
        psgd.update_precond_UVd_math_(U, V, d, v, h, STEP, TINY, balance=False, update_U=(i % 2 == 0))
        psgd.precond_grad_UVd_math(U, V, d, g)
    torch.cuda.synchronize()
