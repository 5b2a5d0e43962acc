import sys, torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd
from tools.kron_bf16_update_timing import tri, timeit
from tools.kron_timing import flops_update
g = torch.Generator(device="cuda"); g.manual_seed(0)
for M in (512, 1000, 1024, 1300, 2048, 2500, 3072):
    for N in (512, 1000, 1300, 2048, 2500, 3072):
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
        dXb, dGb = dX.bfloat16(), dG.bfloat16()
        tb = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01), 10)
        tf = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 10)
        print("%5d %5d  bf16-operand update %7.3f ms  %6.1f TFLOP/s | fp32 update %7.3f ms %6.1f TFLOP/s" % (M, N, tb, flops_update(M, N) / tb * 1e-9, tf, flops_update(M, N) / tf * 1e-9))
