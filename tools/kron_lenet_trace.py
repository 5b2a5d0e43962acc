import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd
from tools.kron_timing import state, LENET
dev = torch.device("cuda:0")
sts = [state(M, N, dev) for M, N in LENET]
Qls, Qrs = [x[0] for x in sts], [x[1] for x in sts]
dXs, dGs, Gs = [x[2] for x in sts], [x[3] for x in sts], [x[4] for x in sts]
for _ in range(5):
    psgd.precond_grad_kron_batched(Qls, Qrs, Gs)
    psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    psgd.precond_grad_kron_batched(Qls, Qrs, Gs)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue time per batched apply: %.1f us; drained after +%.1f us total" % ((t1 - t0) / 200 * 1e6, (t2 - t1) * 1e6))
