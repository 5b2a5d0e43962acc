"""Round 4: the fp32 Kron apply with NEW factors on every call: Gram-free chain (psgd_kron_dd_apply_direct_f32) against prepare + apply
(psgd_kron_dd_apply_f32), and the prepared half alone, over shapes.   python tools/r04_apply_direct_ab.py"""
import sys
import torch
sys.path.insert(0, ".")
from psgd_tf_amd import _lib, kron  # noqa: E402
from tools.kron_bf16_update_timing import tri, timeit  # noqa: E402
lib = _lib.load()
g = torch.Generator(device="cuda"); g.manual_seed(1)
st = torch.cuda.current_stream().cuda_stream
for M, N in ((4096, 4096), (2048, 4096), (4096, 1024), (2048, 2048), (1536, 1536), (1024, 1024), (1100, 520), (640, 2304), (8192, 2048), (6144, 6144)):
    Ql, Qr = tri(M, g), tri(N, g)
    G = torch.randn(M, N, device="cuda", generator=g)
    out = torch.empty_like(G)
    ws = kron._kron_workspace(G.device, M, N)
    a = (Ql.data_ptr(), Qr.data_ptr(), G.data_ptr(), out.data_ptr(), M, N, ws.data_ptr(), ws.numel(), st)
    if lib.psgd_kron_dd_apply_direct_distinct(M, N) != 1:
        print("%dx%d: no direct path" % (M, N)); continue
    t_both = min(timeit(lambda: lib.psgd_kron_dd_apply_f32(*a), 10) for _ in range(2))
    ref = out.clone()
    t_dir = min(timeit(lambda: lib.psgd_kron_dd_apply_direct_f32(*a), 10) for _ in range(2))
    err = float((out - ref).norm() / ref.norm())
    lib.psgd_kron_dd_prepare_f32(Ql.data_ptr(), Qr.data_ptr(), M, N, ws.data_ptr(), ws.numel(), st)
    t_prep = min(timeit(lambda: lib.psgd_kron_dd_apply_prepared_f32(*a), 10) for _ in range(2))
    print("%-10s new factors: prepare + apply %.3f ms, direct %.3f ms (rel diff of the results %.1e); prepared half alone %.3f ms"
          % ("%dx%d" % (M, N), t_both, t_dir, err, t_prep), flush=True)
