"""Per-kernel trace of the small-N (config 2) UVd calls under rocprofv3 (development aid).
    cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -o c2 -- python3 <repo>/tools/c2_trace.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from bench import make_inputs, STEP, TINY  # noqa: E402

from psgd_tf_amd import _lib  # noqa: E402
if os.environ.get("C2_COEF"):  # 1 = block-cooperative reference kernel
    _lib.load().psgd_set_tuning(2, int(os.environ["C2_COEF"]))
if os.environ.get("C2_TPW"):
    _lib.load().psgd_set_tuning(3, int(os.environ["C2_TPW"]))
N, r = int(os.environ.get("C2_N", "1000000")), int(os.environ.get("C2_R", "10"))
dev = torch.device("cuda:0")
U, V, d, g, v, h = make_inputs(N, N, r, dev, 7)
for i in range(60):
    psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, g, STEP, TINY, balance=False, update_U=(i % 2 == 0))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(200):
    psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, g, STEP, TINY, balance=False, update_U=(i % 2 == 0))
e1.record()
torch.cuda.synchronize()
print("fused step wall %.1f us" % (e0.elapsed_time(e1) * 1e3 / 200))
if os.environ.get("C2_UNFUSED"):
    for i in range(60):
        psgd.update_precond_UVd_math_(U, V, d, v, h, STEP, TINY, balance=False, update_U=(i % 2 == 0))
        psgd.precond_grad_UVd_math(U, V, d, g)
    torch.cuda.synchronize()
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, g, STEP, TINY, balance=False, update_U=(i % 2 == 0))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.1f us per call, total %.1f us per call" % ((t1 - t0) * 1e6 / 200, (t2 - t0) * 1e6 / 200))
lib = _lib.load()
ws = psgd.uvd_workspace(dev, N, r)
out = torch.empty_like(g)
st = torch.cuda.current_stream().cuda_stream
P = lambda x: x.data_ptr()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    lib.psgd_uvd_update_apply_f32(P(U), P(V), P(d), P(v), P(h), P(g), P(out), N, r, STEP, TINY, 0, i % 2, P(ws), ws.numel(), st)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("raw C ABI: host enqueue %.1f us per call, total %.1f us per call" % ((t1 - t0) * 1e6 / 200, (t2 - t0) * 1e6 / 200))
