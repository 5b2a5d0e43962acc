"""Round 5: what a one-grid merge of the bf16 apply's two fused triangular pairs could gain at most.  bf16 tuning
key 6 = 1 launches both pairs as ONE grid whose second-pair blocks do not wait for the first pair's tiles (WRONG results, timing only):
pair 2 starts on the CUs pair 1's blocks leave, one launch gap and one ramp are gone -- everything a real merge (which would add a
second hand-off protocol and its waits) could save.   python tools/r05_pair_merge_whatif.py"""
import sys
import torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402
from tools.kron_bf16_update_timing import tri  # noqa: E402
lib = _lib.load()


def timeit(f, n):
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g = torch.Generator(device="cuda"); g.manual_seed(0)
for M, N in ((4096, 4096), (2048, 4096), (4096, 2048), (2560, 2560)):
    Ql, Qr = tri(M, g), tri(N, g)
    G = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    res = []
    for rnd in range(3):
        for key in (0, 1):
            lib.psgd_kron_bf16_set_tuning(6, key)
            res.append((key, timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), 400)))
    lib.psgd_kron_bf16_set_tuning(6, 0)
    t0 = min(t for k, t in res if k == 0); t1 = min(t for k, t in res if k == 1)
    print("%5d x %5d bf16 apply, prepared factors: two launches %.4f ms, one-grid what-if %.4f ms (%.1f %%)" % (M, N, t0, t1, 100 * (t1 - t0) / t0), flush=True)
