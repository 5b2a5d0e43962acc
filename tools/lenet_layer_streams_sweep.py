import sys, torch
sys.path.insert(0, ".")
import preconditioned_stochastic_gradient_descent as psgd
from psgd_tf_amd import kron
LENET5 = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10)]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def tri(n): return torch.triu(torch.randn(n, n, device=dev, generator=g)) * 0.1 + torch.eye(n, device=dev)
sts = [(tri(m), tri(n), torch.randn(m, n, device=dev, generator=g)) for m, n in LENET5]
dXs = [torch.randn_like(s[2]) for s in sts]
def timeit(f, n=200):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def graphed(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side): fn()
    return gr.replay
for n in (1, 2, 3, 5, 8):
    def ap():
        with kron.layer_streams(n): return [psgd.precond_grad_kron(a, b, c) for a, b, c in sts]
    def up():
        with kron.layer_streams(n): return [psgd.update_precond_kron(a, b, x, c, 0.01) for (a, b, c), x in zip(sts, dXs)]
    print("streams %d: graph replay apply %.1f us  update %.1f us" % (n, timeit(graphed(ap)), timeit(graphed(up))), flush=True)
# order: largest layers first
order = sorted(range(5), key=lambda i: -LENET5[i][0] * LENET5[i][1])
def ap2():
    with kron.layer_streams(5): return [psgd.precond_grad_kron(*sts[i]) for i in order]
def up2():
    with kron.layer_streams(5): return [psgd.update_precond_kron(sts[i][0], sts[i][1], dXs[i], sts[i][2], 0.01) for i in order]
print("streams 5, largest first: apply %.1f us  update %.1f us" % (timeit(graphed(ap2)), timeit(graphed(up2))))
