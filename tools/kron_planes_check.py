"""Large fp32 Kron apply on pre-split planes (tuning key 4) against the in-GEMM split and an fp64 reference; timings.
    python tools/kron_planes_check.py"""
import sys
import torch

sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402

if __name__ == "__main__":
    lib = _lib.load()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    shapes = [(1024, 1024), (1100, 1030), (1024, 2049), (2049, 1024), (1500, 1027), (4096, 4096), (2048, 8192), (8192, 2048)]
    if len(sys.argv) > 1:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    for M, N in shapes:
        Ql, Qr = tri(M, g), tri(N, g)
        G = torch.randn(M, N, device="cuda", generator=g)
        ref = (Ql.double().T @ Ql.double()) @ G.double() @ (Qr.double().T @ Qr.double())
        res = {}
        for planes in (0, 1):
            lib.psgd_kron_set_tuning(4, planes)
            Ql2, Qr2 = Ql.clone(), Qr.clone()                     # new factor tensors: the prepared state is rebuilt
            out = psgd.precond_grad_kron(Ql2, Qr2, G)
            err = ((out.double() - ref).norm() / ref.norm()).item()
            t_cached = timeit(lambda: psgd.precond_grad_kron(Ql2, Qr2, G), 10)
            def cold():
                Ql2.add_(0.0)                                      # bumps the version: factors count as changed
                return psgd.precond_grad_kron(Ql2, Qr2, G)
            t_cold = timeit(cold, 10)
            res[planes] = out
            print("Kron fp32 apply %5dx%-5d planes=%d: rel err vs fp64 %.2e, unchanged factors %.3f ms, new factors %.3f ms" % (M, N, planes, err, t_cached, t_cold))
        d = ((res[0] - res[1]).abs().max() / res[0].abs().max()).item()
        print("      max |planes - in-GEMM split| / max|out| = %.2e" % d)
    lib.psgd_kron_set_tuning(4, 1)
