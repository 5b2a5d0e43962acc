"""Quick per-stage timing of the UVd path on one GPU (development aid, not the bench contract)."""
import argparse
import time

import torch

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=100_000_000)
    ap.add_argument("--r", type=int, default=20)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--bpc", type=int, default=0, help="cap on blocks per CU (tuning key 1)")
    ap.add_argument("--alias-shift", type=int, default=-1,
                    help="experiment: V = U's buffer shifted by this many rows (same DRAM region)")
    ap.add_argument("--swap-alloc", action="store_true", help="experiment: allocate V before U")
    ap.add_argument("--pad-mb", type=int, default=0, help="experiment: dummy allocation between U and V (MiB)")
    ap.add_argument("--pad-ws-kb", type=int, default=0, help="experiment: dummy allocation before the workspace (KiB)")
    ap.add_argument("--only", default="", help="comma-separated stage names to time")
    args = ap.parse_args()
    N, r = args.N, args.r
    dev = torch.device("cuda:0")
    lib = _lib.load()
    if args.bpc:
        lib.psgd_set_tuning(1, args.bpc)
    g = torch.Generator(device=dev).manual_seed(0)
    scale = (1.0 / (N * r)) ** 0.5
    if args.swap_alloc:
        V0 = torch.randn(N, r, device=dev, generator=g) * scale
    U = torch.randn(N, r, device=dev, generator=g) * scale
    if args.pad_mb:
        pad = torch.empty(args.pad_mb * 2**20, dtype=torch.uint8, device=dev)
    if args.swap_alloc:
        V = V0
    elif args.alias_shift >= 0:
        buf = torch.randn(N + args.alias_shift, r, device=dev, generator=g) * scale
        U, V = buf[:N], buf[args.alias_shift:]
    else:
        V = torch.randn(N, r, device=dev, generator=g) * scale
    d = torch.ones(N, 1, device=dev)
    gr = torch.randn(N, 1, device=dev, generator=g)
    v = torch.randn(N, 1, device=dev, generator=g)
    h = v * torch.exp(torch.empty(N, 1, device=dev).uniform_(-4.6, 4.6, generator=g))
    if args.pad_ws_kb:
        pad2 = torch.empty(args.pad_ws_kb * 1024, dtype=torch.uint8, device=dev)
    ws = psgd.uvd_workspace(dev, N, r)
    out = torch.empty_like(gr)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()

    stages = {
        "apply_s1": lambda: lib.psgd_uvd_apply_sweep1_f32(P(V), P(d), P(gr), N, r, P(ws), ws.numel(), st),
        "apply_s2": lambda: lib.psgd_uvd_apply_sweep2_f32(P(U), P(d), P(gr), P(out), N, r, 0, P(ws), ws.numel(), st),
        "apply_s3": lambda: lib.psgd_uvd_apply_sweep3_f32(P(V), P(d), P(out), N, r, 0, P(ws), ws.numel(), st),
        "apply": lambda: lib.psgd_uvd_apply_f32(P(U), P(V), P(d), P(gr), P(out), N, r, P(ws), ws.numel(), st),
        "upd_s1": lambda: lib.psgd_uvd_update_sweep1_f32(P(U), P(V), P(d), P(v), P(h), N, r, P(ws), ws.numel(), st),
        "upd_s2U": lambda: lib.psgd_uvd_update_sweep2_f32(P(U), P(V), P(d), P(v), P(h), N, r, 0.01, 1.1754944e-38, 1, P(ws), ws.numel(), st),
        "upd_s2V": lambda: lib.psgd_uvd_update_sweep2_f32(P(U), P(V), P(d), P(v), P(h), N, r, 0.01, 1.1754944e-38, 0, P(ws), ws.numel(), st),
        "upd_s3": lambda: lib.psgd_uvd_update_sweep3_f32(P(d), N, r, 0.01, 1.1754944e-38, P(ws), ws.numel(), st),
        "update": lambda: lib.psgd_uvd_update_f32(P(U), P(V), P(d), P(v), P(h), N, r, 0.01, 1.1754944e-38, 0, 1, P(ws), ws.numel(), st),
    }
    bytes_per_row = {
        "apply_s1": 4 * (r + 2), "apply_s2": 4 * (r + 3), "apply_s3": 4 * (r + 3), "apply": 4 * (4 * r + 5),
        "upd_s1": 4 * (2 * r + 3), "upd_s2U": 4 * (3 * r + 4), "upd_s2V": 4 * (3 * r + 4), "upd_s3": 12,
        "update": 4 * (5 * r + 10),
    }
    print("U %x V %x d %x ws %x" % (U.data_ptr(), V.data_ptr(), d.data_ptr(), ws.data_ptr()))
    only = set(x for x in args.only.split(",") if x)
    for name, fn in stages.items():
        if only and name not in only:
            continue
        for _ in range(2):
            rc = fn()
            assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        gbs = bytes_per_row[name] * N / (ms * 1e-3) / 1e9
        print("%-9s %9.3f ms  %8.1f GB/s algorithmic  (%.1f%% of 8 TB/s)  %.2f Gparam/s" %
              (name, ms, gbs, gbs / 80.0, N / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
