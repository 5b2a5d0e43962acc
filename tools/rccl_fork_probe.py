"""Does an initialised RCCL communicator in the process change the two-stream Kron update?  (Round 3: bench.py's
exchange_overhead leg -- a 1-rank RCCL group, created and destroyed -- ran before the Kron leg, and the 4096^2 updates
came out 1.1-1.3 ms slower than without it.)  Times the 4096^2 fp32 and bf16-operand update with the chains forked
(tuning key 9 = 1) and serial (0), before any process group, with a live 1-rank group, and after destroying it.
   python tools/rccl_fork_probe.py [side-stream priority: 0 lowest (default) | 1 default | 2 highest] [late]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import _lib
from tools.kron_timing import state

lib = _lib.load()
prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib.psgd_kron_set_tuning(10, prio)
dev = torch.device("cuda:0")
M = N = 4096
Ql, Qr, dX, dG, G = state(M, N, dev)
dXb, dGb = dX.bfloat16(), dG.bfloat16()


def t_of(fn, n=10):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


mid = {n: state(n, n, dev) for n in (1024, 2048)}       # sizes where the fork is worth 1.2-1.35x


def row(tag):
    out = []
    for ov in (1, 0):
        lib.psgd_kron_set_tuning(9, ov)
        out.append(t_of(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)))
        out.append(t_of(lambda: psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01)))
        for n in (1024, 2048):
            a, b, x, g_, _ = mid[n]
            out.append(t_of(lambda: psgd.update_precond_kron(a, b, x, g_, 0.01), 40))
    lib.psgd_kron_set_tuning(9, 1)
    print("%-28s forked: 4096 f32 %.3f bf16 %.3f | 1024 %.3f | 2048 %.3f ms   serial: 4096 f32 %.3f bf16 %.3f | 1024 %.3f | 2048 %.3f ms"
          % (tag, *out), flush=True)


late = len(sys.argv) > 2 and sys.argv[2] == "late"      # first forked call (= side stream creation) only after the group is gone
print("side-stream priority mode %d%s" % (prio, " (first update after the process group)" if late else ""))
if late:
    _row = row
    row = lambda tag: print("%-28s (skipped)" % tag) if tag != "group destroyed" else _row(tag)
row("no process group")
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
row("group initialised (lazy)")
x = torch.ones(8, device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
row("after one all_reduce")
# what bench.py's exchange_overhead leg does between creating and destroying the group: the fused UVd step at 12.5M rows,
# unsharded and through psgd_tf_amd/sharded.py (2 all-gathers + 2 fold kernels per step)
from psgd_tf_amd import sharded
import bench
U, V, d, grad, v, h = bench.make_inputs(12_500_032, 12_500_032, 20, dev, 0)
for i in range(50):
    psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, grad, 0.01, bench.TINY, balance=False, update_U=(i % 2 == 0))
torch.cuda.synchronize()
row("after 50 unsharded UVd steps")
for i in range(50):
    sharded.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, grad, 0.01, bench.TINY, balance=False, update_U=(i % 2 == 0))
torch.cuda.synchronize()
row("after 50 sharded UVd steps")
del U, V, d, grad, v, h
torch.cuda.empty_cache()
row("after empty_cache")
dist.destroy_process_group()
row("group destroyed")
