#!/usr/bin/env python3
"""bench.py -- headline benchmark of the PSGD preconditioner hot path on MI355X.

Metric (BASELINE.json): params/sec for the UVd update + apply at N = 100M rows, r = 20:
one "step" = update_precond_UVd_math_(U,V,d,v,h,step,tiny) followed by
precond_grad_UVd_math(U,V,d,g) (the UVd.step call pattern, psgd.py:732 -> :748) on synthetic
(g, v, h) already resident in HBM.

  python bench.py --gpus N --steps K --warmup W

N > 1 started from a plain shell launches its own ranks: the parent process (which never touches a GPU) starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` as a child, relays
rank 0's JSON line and exits with the child's code.  Started under torch.distributed.run (WORLD_SIZE set) it is
one of the ranks.  N > 1 row-shards the flat parameter vector (psgd_tf_amd/sharded.py); per step two tiny
all-gathers (RCCL over xGMI) carry the r-dimensional reduced buffers -- never N-sized data.  Which rows:
  (default)        BASELINE configs[3]: N_global = 100M rows, r = 20 split over the N ranks in contiguous row blocks
                   (12.5M rows per GPU at N = 8) -> "scaling": "strong"; the line also carries a `weak` sub-record
                   (100M rows PER GPU, global N = 100M * world) measured right after it (--no-weak-leg skips it)
  --global-rows G  strong scaling at G global rows only
  --rows R         weak scaling only: every rank holds R rows
At N = 1 both defaults are the same workload (100M rows on the one GPU = BASELINE's metric).  The N = 1 line also
carries `exchange_overhead`: one rank's share of configs[3] (12.5M rows) run unsharded and through the multi-GPU
path on a 1-rank RCCL group -- the us per step the two exchanges + fold kernels add (--no-exchange-leg skips it).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel (update sweep 2), timed live with HIP events on the launch stream
                  (psgd_prof_*) in a SECOND pass (the headline pass runs with the hooks off); algorithmic bytes
                  per launch / average duration vs 8 TB/s.  `paths` holds the whole fused step and, at N = 1,
                  separate legs for precond_grad_UVd_math alone and update_precond_UVd_math_ alone (SURVEY 8d
                  byte counts and the bytes the sweeps actually move), and the same three at config 2
                  (N = 1M, r = 10).
  cpu_baseline -- the torch-CPU restatement of the reference op sequence (oracle/, "port") timed on this host's
                  cores (N = 1 run only): thread-count sweep, then 2 warm-up + 5 timed steps, median, on a
                  bounded sample of the same workload (--cpu-full: the full N if host memory allows).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TINY = 1.1754943508222875e-38  # psgd.py:22
STEP = 0.01                    # psgd.py:664 (lr_preconditioner default)


def make_inputs(n_local, n_global, r, dev, seed, arena=None):
    """SURVEY 8d synthetic inputs: U,V ~ N(0,1)(N r)^-1/2 (psgd.py:687-689), d = 1 (:690),
    g,v ~ N(0,1) (:713), h = c.*v with c ~ LogUniform[1e-2,1e2].
    arena: a psgd_tf_amd.placement.UVdArena whose regions receive the same values (the state's owner decides where it lives)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    scale = (1.0 / (n_global * r)) ** 0.5
    g1 = torch.Generator(device=dev).manual_seed(seed + 1)
    if arena is not None:
        U, V, d, grad, v, h = arena.U, arena.V, arena.d, arena.g, arena.v, arena.h
        torch.randn(n_local, r, device=dev, generator=g, out=U).mul_(scale)
        torch.randn(n_local, r, device=dev, generator=g, out=V).mul_(scale)
        d.fill_(1.0)
        torch.randn(n_local, 1, device=dev, generator=g1, out=grad)
        torch.randn(n_local, 1, device=dev, generator=g1, out=v)
        h.uniform_(-4.605170186, 4.605170186, generator=g1).exp_().mul_(v)
        return U, V, d, grad, v, h
    U = torch.randn(n_local, r, device=dev, generator=g) * scale
    V = torch.randn(n_local, r, device=dev, generator=g) * scale
    d = torch.ones(n_local, 1, device=dev)
    grad = torch.randn(n_local, 1, device=dev, generator=g1)
    v = torch.randn(n_local, 1, device=dev, generator=g1)
    c = torch.exp(torch.empty(n_local, 1, device=dev).uniform_(-4.605170186, 4.605170186, generator=g1))
    h = c * v
    return U, V, d, grad, v, h


def _host_mem_available_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 2**20
    except OSError:
        pass
    return None


def cpu_baseline(r, sample_rows, budget_s, threads=None, full_rows=None):
    """Reference op sequence on torch-CPU (oracle/psgd_oracle_torch.py), update + apply.
    1. thread-count sweep on the first 1M rows (skinny [N, r] matmuls do not scale to every core), the best three counts timed again
       on the whole `sample_rows` sample (a bandwidth-bound size can want more threads than a cache-sized one);
    2. with the best count 2 warm-up + up to 5 timed steps on the sample, median  -> `sample_value`;
    3. full_rows (the metric's N, given when the host has the memory): 1 warm-up + 2 timed steps on the FULL workload -> `value`
       (same box, full config); without it `value` is the sample's figure and `full_n_skipped` says why."""
    import statistics
    from oracle import psgd_oracle_torch as ref
    ncpu = os.cpu_count() or 1

    def stepper(t):
        U, V, d, grad, v, h = t

        def one(i):
            t0 = time.perf_counter()
            ref.update_precond_UVd_math_(U, V, d, v, h, STEP, TINY, balance=False, update_U=(i % 2 == 0))
            t1 = time.perf_counter()
            ref.precond_grad_UVd_math(U, V, d, grad)
            t2 = time.perf_counter()
            return t2 - t0, t1 - t0, t2 - t1
        return one

    data = make_inputs(sample_rows, sample_rows, r, torch.device("cpu"), 0)
    one = stepper(data)
    sweep_rows = min(sample_rows, 1_000_000)
    one_small = stepper(tuple(x[:sweep_rows].clone() for x in data))
    t_start = time.perf_counter()
    cands = [threads] if threads else sorted({t for t in (4, 8, 16, 32, 64, 128, ncpu) if t <= ncpu})
    sweep = {}
    for t in cands:
        torch.set_num_threads(t)
        one_small(0)
        sweep[t] = min(one_small(1)[0], one_small(2)[0])
        if time.perf_counter() - t_start > 0.4 * budget_s:
            break
    resweep = {}
    for t in sorted(sweep, key=sweep.get)[:3]:                 # the best three again, on the timed size
        torch.set_num_threads(t)
        one(0)
        resweep[t] = min(one(1)[0], one(2)[0])
        if time.perf_counter() - t_start > 0.9 * budget_s and resweep:
            break
    best = min(resweep, key=resweep.get)
    torch.set_num_threads(best)
    for i in range(2):
        one(i)
    runs = []
    for i in range(5):
        runs.append(one(i))
        if time.perf_counter() - t_start > 2.0 * budget_s and len(runs) >= 3:
            break
    med = statistics.median(x[0] for x in runs)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    del data, one, one_small
    out = {"value": sample_rows / med, "unit": "params/s", "cores": best, "kind": "port",
           "sample": "update+apply on N=%d rows (of the metric's 100M), r=%d fp32: %d timed steps, median %.3f s; %d threads"
                     % (sample_rows, r, len(runs), med, best),
           "sample_detail": "torch-CPU restatement of psgd.py:554-627 in the reference's op order (TensorFlow unavailable), "
                            "2 warm-up steps, min %.3f s; thread sweep on %d rows (s per step): %s; best three on %d rows: %s; "
                            "host: %d logical CPUs, %s"
                            % (min(x[0] for x in runs), sweep_rows, {k: round(v_, 3) for k, v_ in sweep.items()}, sample_rows,
                               {k: round(v_, 3) for k, v_ in resweep.items()}, ncpu, model),
           "sample_value": sample_rows / med, "sample_rows": sample_rows,
           "update_params_per_s": sample_rows / statistics.median(x[1] for x in runs),
           "apply_params_per_s": sample_rows / statistics.median(x[2] for x in runs),
           "thread_sweep_max": max(sweep), "logical_cpus": ncpu, "cpu_model": model,
           "host_mem_available_gib": _host_mem_available_gib(), "full_n_value": None, "full_n_skipped": None}
    if full_rows and full_rows > sample_rows:
        need_gib = full_rows * (8 * r + 24 + 3 * 4 * r) * 1.25 / 2**30        # U, V, six vectors, three [N, r] temporaries, slack
        avail = out["host_mem_available_gib"]
        if avail is None or avail < need_gib + 8.0:
            out["full_n_skipped"] = "host MemAvailable %.0f GiB < %.0f GiB needed for N=%d, r=%d" % (avail or 0, need_gib + 8.0,
                                                                                                  full_rows, r)
        else:
            one = stepper(make_inputs(full_rows, full_rows, r, torch.device("cpu"), 0))
            one(0)
            fr = [one(1), one(2)]
            fmed = statistics.median(x[0] for x in fr)
            out.update(value=full_rows / fmed, full_n_value=full_rows / fmed, full_n_rows=full_rows,
                       sample="update+apply on the metric's FULL N=%d rows, r=%d fp32, this host: 1 warm-up + 2 timed steps, "
                              "%.2f / %.2f s; %d threads (chosen on a %d-row sample: %.0f params/s there)"
                              % (full_rows, r, fr[0][0], fr[1][0], best, sample_rows, sample_rows / med),
                       update_params_per_s=full_rows / statistics.median(x[1] for x in fr),
                       apply_params_per_s=full_rows / statistics.median(x[2] for x in fr))
    elif not full_rows:
        out["full_n_skipped"] = "--cpu-sample-only"
    return out


LENET5 = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10)]      # mnist_with_lenet5.py:12-16


def kron_apply_flops(M, N):
    """F_ref of SURVEY 8d: the dense flops of the reference's op sequence (psgd.py:189-192)."""
    return 2 * M**3 + 2 * M * M * N + 4 * M * N * N if M < N else 2 * N**3 + 2 * M * N * N + 4 * M * M * N


def kron_bench(dev, psgd, iters=20):
    """Second half of the BASELINE metric: Kron dense(x)dense apply GFLOP/s (F_ref numerator).
    4096 x 4096 with bf16 MFMA operands (config 5), the same in exact fp32, and the LeNet5 layer
    set in fp32 (config 3; launch/latency-bound: reported as us per set)."""
    def state(M, N):
        g = torch.Generator(device=dev).manual_seed(M * 7 + N)
        Ql = torch.triu(torch.randn(M, M, device=dev, generator=g) * 0.02, 1) + torch.eye(M, device=dev)
        Qr = torch.triu(torch.randn(N, N, device=dev, generator=g) * 0.02, 1) + torch.eye(N, device=dev)
        # ten warm-up updates' worth of structure is not needed for timing: factors are dense upper-triangular
        return Ql, Qr, torch.randn(M, N, device=dev, generator=g)

    def timeit(fn, n, warm_ms=60.0, min_ms=20.0):
        # Steady-state clocks: the same call runs for >= 60 ms before the timed region, which covers >= 20 ms.  After an
        # idle gap (tensor set-up, a host sync) the device takes tens of ms to settle its clock: the same 0.5-ms GEMM
        # measures 0.52 or 0.65 ms depending on what ran in the milliseconds before it (profiles/r02_kron_x3_whatif.txt).
        # (round 6: the warm-up is counted in MEASURED device time, in chunks of ~10 ms, and the call's duration is re-estimated from the
        #  last chunk -- the first three calls after a gap run at 0.4-1 ms where the settled call takes 0.26, so a warm-up sized from them
        #  was 10-20 ms and the timed region still sat on the clock ramp: the first leg of this function read 0.29-0.33 ms for a call that
        #  every stand-alone run of the same protocol measures at 0.262)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        spent, per, chunk = 0.0, 1.0, 3
        while spent < warm_ms:
            e0.record()
            for _ in range(chunk):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            dt = e0.elapsed_time(e1)
            spent += dt
            per = max(dt / chunk, 1e-3)
            chunk = min(2000, max(3, int(10.0 / per)))
        n = max(n, min(2000, int(min_ms / per) + 1))
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n

    M = N = 4096
    Ql, Qr, G = state(M, N)
    Gb = G.to(torch.bfloat16)
    t_bf16 = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, Gb), iters)
    Ql2, Qr2 = Ql.clone(), Qr.clone()
    pairs = [(Ql, Qr), (Ql2, Qr2)]
    flip = [0]

    def cold():                                   # alternate two factor pairs: the cached bf16 copies never match
        flip[0] ^= 1
        return psgd.precond_grad_kron(pairs[flip[0]][0], pairs[flip[0]][1], Gb)
    t_bf16_cold = timeit(cold, iters)
    del Ql2, Qr2
    t_f32 = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), iters)
    Ql2, Qr2 = Ql.clone(), Qr.clone()
    pairs = [(Ql, Qr), (Ql2, Qr2)]
    flip = [0]

    def cold32():                                 # alternate two factor pairs: the prepared Gram never matches
        flip[0] ^= 1
        return psgd.precond_grad_kron(pairs[flip[0]][0], pairs[flip[0]][1], G)
    t_f32_cold = timeit(cold32, iters)
    from psgd_tf_amd import kron as _kron
    old_route = _kron.set_apply_route("auto")      # the opt-in fast path: new factors take the Gram-free chain
    try:
        t_f32_cold_auto = timeit(cold32, iters)
    finally:
        _kron.set_apply_route(old_route)
    del Ql2, Qr2, pairs
    # the large updates BEFORE the LeNet5 legs: those create a pool of streams and captured graphs, and a process that holds many streams
    # maps the update's caller stream and its side stream onto shared hardware queues -- the two chains of the update then overlap less
    # (measured: 2.24 ms alone, 2.43 ms at the end of this function)
    dX = torch.randn_like(G)
    t_upd = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, G, 0.01), 5)
    dXb, dGb = dX.to(torch.bfloat16), Gb
    t_upd_bf16 = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dXb, dGb, 0.01), 5)
    # the same fp32 calls on the EXACT fp32 matrix core (v_mfma_f32_16x16x4_f32; tuning key 1 = 0): the default large-layer path
    # multiplies f16 x 2 planes (about 22-bit operands, fp32 accumulation: 5.7e-7 against fp64 at 4096^2), not IEEE fp32 products
    t_f32_exact = t_upd_exact = None
    try:
        _kron.set_tuning(1, 0)
        t_f32_exact = timeit(lambda: psgd.precond_grad_kron(Ql, Qr, G), 3)
        t_upd_exact = timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, G, 0.01), 3)
    except Exception as exc:
        print("exact-fp32 Kron leg failed: %r" % (exc,), file=sys.stderr)
    finally:
        _kron.set_tuning(1, 1)
    del dX, dXb
    sts = [state(m, n) for m, n in LENET5]
    Qls, Qrs, Gs = [x[0] for x in sts], [x[1] for x in sts], [x[2] for x in sts]
    t_lenet = timeit(lambda: psgd.precond_grad_kron_batched(Qls, Qrs, Gs), 50)        # one launch per stage for all layers
    t_lenet_loop = timeit(lambda: [psgd.precond_grad_kron(a, b, c) for a, b, c in sts], 50)
    sts2 = [(a.clone(), b.clone(), c) for a, b, c in sts]
    both = [sts, sts2]

    def cold_loop():                              # per-layer calls with factors that changed since the last call
        flip[0] ^= 1
        return [psgd.precond_grad_kron(a, b, c) for a, b, c in both[flip[0]]]

    def cold_batched():
        flip[0] ^= 1
        cur = both[flip[0]]
        return psgd.precond_grad_kron_batched([x[0] for x in cur], [x[1] for x in cur], Gs)
    t_lenet_loop_cold = timeit(cold_loop, 50)
    t_lenet_cold = timeit(cold_batched, 50)
    dXs = [torch.randn_like(g_) for g_ in Gs]
    t_lenet_upd = timeit(lambda: psgd.update_precond_kron_batched(Qls, Qrs, dXs, Gs, 0.01), 50)
    t_lenet_upd_loop = timeit(lambda: [psgd.update_precond_kron(a, b, x, g, 0.01)          # mnist_with_lenet5.py:51
                                       for a, b, x, g in zip(Qls, Qrs, dXs, Gs)], 50)

    def graphed(fn):
        """The reference's per-layer list comprehension captured ONCE in a graph and replayed: the drop-in route that removes the
        host cost of five Python calls per step (capture is legal for both entry points: tests/test_kron_gpu.py)."""
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()                                  # workspaces of the capture stream exist before the capture
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            fn()
        return gr.replay
    try:
        t_lenet_loop_graph = timeit(graphed(lambda: [psgd.precond_grad_kron(a, b, c) for a, b, c in sts]), 50)       # :53
        t_lenet_upd_loop_graph = timeit(graphed(lambda: [psgd.update_precond_kron(a, b, x, g, 0.01)                  # :51
                                                         for a, b, x, g in zip(Qls, Qrs, dXs, Gs)]), 50)
    except Exception as exc:                          # a graph failure must not take the bench line down
        print("lenet graph leg failed: %r" % (exc,), file=sys.stderr)
        t_lenet_loop_graph = t_lenet_upd_loop_graph = float("nan")
    # the same per-layer calls, each on its own forked stream (`with kron.layer_streams():` around the list comprehension: the
    # layers are independent and each call is a chain of 3-5 dependent launches); eager, and as a graph with one branch per layer
    def forked_apply():                           # (new factors on every call, like cold_loop; inside a capture the Grams are always rebuilt)
        flip[0] ^= 1
        with _kron.layer_streams():
            return [psgd.precond_grad_kron(a, b, c) for a, b, c in both[flip[0]]]

    def forked_update():
        with _kron.layer_streams():
            return [psgd.update_precond_kron(a, b, x, g, 0.01) for a, b, x, g in zip(Qls, Qrs, dXs, Gs)]
    t_lenet_forked = t_lenet_upd_forked = t_lenet_forked_graph = t_lenet_upd_forked_graph = float("nan")
    try:
        t_lenet_forked = timeit(forked_apply, 50)
        t_lenet_upd_forked = timeit(forked_update, 50)
        t_lenet_forked_graph = timeit(graphed(forked_apply), 50)
        t_lenet_upd_forked_graph = timeit(graphed(forked_update), 50)
    except Exception as exc:
        print("lenet forked-streams leg failed: %r" % (exc,), file=sys.stderr)
    # ... and inside `with kron.layer_batch():` -- the calls only queue, leaving the block issues ONE batched launch sequence per kind
    def lb_apply():                               # (new factors on every call, like cold_loop)
        flip[0] ^= 1
        with _kron.layer_batch():
            return [psgd.precond_grad_kron(a, b, c) for a, b, c in both[flip[0]]]

    def lb_update():
        with _kron.layer_batch():
            return [psgd.update_precond_kron(a, b, x, g, 0.01) for a, b, x, g in zip(Qls, Qrs, dXs, Gs)]
    t_lb = t_lb_upd = t_lb_graph = t_lb_upd_graph = float("nan")
    try:
        t_lb = timeit(lb_apply, 50)
        t_lb_upd = timeit(lb_update, 50)
        t_lb_graph = timeit(graphed(lb_apply), 50)
        t_lb_upd_graph = timeit(graphed(lb_update), 50)
    except Exception as exc:
        print("lenet layer_batch leg failed: %r" % (exc,), file=sys.stderr)
    f_upd = 7 * (M * M * N + M * N * N) + 2 * (M**3 + N**3)                        # SURVEY 8d F_ref of the update
    f_big = kron_apply_flops(M, N)
    f_lenet = sum(kron_apply_flops(m, n) for m, n in LENET5)
    # flops the bf16 apply ISSUES: two fused triangular pairs on 256^2 tiles; a tile in tile row i of a pair runs
    # (T - i) + (i + 1) = T + 1 K chunks of 256 (T = tile rows of the triangular factor), 2 * 256^3 flops each
    # (matches rocprofv3's MOPS_BF16 count, profiles/kron_mfma_pmc.json: 292 GFLOP at 4096^2)
    tm, tn = M // 256, N // 256
    f_issued = (tn * tm * (tn + 1) + tm * tn * (tm + 1)) * 2 * 256**3
    # fp32 apply on planes (M >= N order, psgd_kron.hip planes_apply): tile steps of G (Qr'Qr), Ql (.), Ql' (.)
    t128m, t128n = -(-M // 128), -(-N // 128)
    steps = t128m * t128n * -(-N // 32)                                                   # full K = N
    steps += sum(t128n * -(-(M - 128 * i) // 32) for i in range(t128m))                    # K from the tile row on
    steps += sum(t128n * -(-min(M, 128 * (i + 1)) // 32) for i in range(t128m))            # K up to the tile row
    f32_issued = steps * 3 * 2 * 128 * 128 * 32                                           # f16 x 2 planes: 3 MFMAs per term
    pmc = None                          # matrix-core counters of the same call, collected with rocprofv3 --pmc
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "kron_mfma_pmc.json")))
        pmc = {"source": "profiles/kron_mfma_pmc.json (rocprofv3 --pmc on an earlier box, NOT counters of this run)",
               "issued_gflop_per_apply": pmc["issued_flops_per_apply"] / 1e9,
               "MfmaUtil_percent": {k: v["mfma_util_percent"] for k, v in pmc["kernels"].items()},
               "MfmaUtil_percent_time_weighted": pmc["mfma_util_percent_time_weighted"]}
    except Exception:
        pmc = None
    # Headline times are the NEW-FACTORS-EVERY-CALL ones: in the reference's call pattern the factors change before
    # every apply (mnist_with_lenet5.py:51 -> :53), so the bf16 factor copies / the Gram of psgd.py:192 are made inside
    # the timed call and every flop F_ref counts is executed (or skipped as a triangular zero block) in that time.
    # `*_unchanged_factors` keep the prepared state across calls (factors untouched between applies).
    return {
        "metric": "kron_dense_dense_apply_gflops", "flop_count": "F_ref (dense flops of psgd.py:189-192)",
        "timing": "every `ms` / `us` / `gflops` below is measured with NEW factors on every call (nothing cached "
                  "between calls: the reference's update -> apply pattern); `*_unchanged_factors` reuse the prepared "
                  "factor state",
        "roofline": {"bound": "mfma", "kernel": "k_hgemm_tri_pair_256 x 2 + factor conversion (bf16 apply, 4096 x 4096)",
                     "achieved": f_issued / t_bf16_cold / 1e9, "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": f_issued / t_bf16_cold / 1e9 / 2500.0, "issued_gflop_per_apply": f_issued / 1e9,
                     "F_ref_gflop_per_apply": f_big / 1e9, "ms": t_bf16_cold,
                     "frac_unchanged_factors": f_issued / t_bf16 / 1e9 / 2500.0,
                     "note": "fraction of the dense bf16 MFMA peak on the flops the kernels ISSUE (triangular zero blocks are "
                             "skipped: ~0.53 of F_ref), whole call incl. the fp32 -> bf16 factor conversion and the flag memset, "
                             "new factors every call"},
        "4096x4096_bf16_operands": {"ms": t_bf16_cold, "gflops": f_big / t_bf16_cold / 1e6, "mfma_peak_gflops": 2.5e6,
                                    "frac_of_bf16_peak_Fref": f_big / t_bf16_cold / 1e6 / 2.5e6,
                                    "frac_of_bf16_peak_issued": f_issued / t_bf16_cold / 1e6 / 2.5e6,
                                    "ms_unchanged_factors": t_bf16, "gflops_unchanged_factors": f_big / t_bf16 / 1e6,
                                    "note": "triangular K-ranges skipped: issued flops ~0.53 F_ref; `ms` rebuilds the bf16 factor "
                                            "copies every call (a fresh factor pair per call, as right after an update), "
                                            "`ms_unchanged_factors` reuses them",
                                    "mfma_pmc": pmc},
        "4096x4096_fp32": {"ms": t_f32_cold, "gflops": f_big / t_f32_cold / 1e6, "mfma_peak_gflops": 157.3e3,
                           "arithmetic": "f16x2-plane fp32 emulation (fp32 accumulation)", "route": "reference (default): Gram of psgd.py:192 every call",
                           "ms_auto_route": t_f32_cold_auto, "ms_exact_fp32_mfma": t_f32_exact,
                           "ms_unchanged_factors": t_f32, "gflops_unchanged_factors": f_big / t_f32 / 1e6,
                           "issued_f16_gflop_per_apply_unchanged_factors": f32_issued / 1e9,
                           "frac_of_f16_peak_issued_unchanged_factors": f32_issued / t_f32 / 1e6 / 2.5e6,
                           "note": "fp32-accurate products on the fp16 matrix cores (same dense peak as bf16): operands split once "
                                   "into two fp16 planes and a power-of-two scale per matrix (x 2^e = h + 2^-11 M), 3 MFMAs per "
                                   "product term, K loop = DMA + MFMA (k_gemm_p3<1>); `issued` counts the 128 x 128 x 32 tile steps "
                                   "the three gradient-side products of the prepared form run (triangular K ranges skipped) x 3; `ms` = new factors on "
                                   "every call: no Gram, out = Ql'(Ql((G Qr')Qr)) as four chained triangular products with the factors' "
                                   "planes made inside the call; `ms_unchanged_factors` = the prepared form (Gram of psgd.py:192 and factor "
                                   "planes kept from the second call with the same factor tensors on); the "
                                   "fp32 MFMA peak is quoted for reference, it does not bound this kernel"},
        "lenet5_set_fp32": {"us": t_lenet_loop_cold * 1e3, "gflops": f_lenet / t_lenet_loop_cold / 1e6, "bound": "launch/latency",
                            "call": "[precond_grad_kron(Ql, Qr, G) for each layer]  (the reference's bare pattern, mnist_with_lenet5.py:53, "
                                    "new factors every call); `layer_batch_us` = the same comprehension inside `with kron.layer_batch():`",
                            "per_layer_calls_us": t_lenet_loop_cold * 1e3,
                            "layer_batch_us": t_lb * 1e3, "layer_batch_update_us": t_lb_upd * 1e3,
                            "layer_batch": {"apply_us": t_lb * 1e3, "apply_graph_us": t_lb_graph * 1e3,
                                            "update_us": t_lb_upd * 1e3,
                                            "update_graph_us": t_lb_upd_graph * 1e3,
                                            "ratio_to_batched": t_lb / t_lenet_cold, "update_ratio_to_batched": t_lb_upd / t_lenet_upd},
                            "per_layer_calls_unchanged_factors_us": t_lenet_loop * 1e3,
                            "per_layer_calls_graph_us": t_lenet_loop_graph * 1e3,
                            "batched_us": t_lenet_cold * 1e3, "batched_us_unchanged_factors": t_lenet * 1e3,
                            "batched_call": "precond_grad_kron_batched (extension: one launch per stage for all layers)",
                            "note": "`us` = `per_layer_calls_us`: five per-layer calls, new factors on every call (3 launches each); "
                                    "`*_unchanged_factors*`: the Grams stay prepared (2 launches); `*_graph_us`: the same list "
                                    "comprehension captured once in a CUDA graph and replayed (no host cost; inside a capture the "
                                    "Grams are always rebuilt); `batched_us`: the batched extension, new factors",
                            "update_us": t_lenet_upd_loop * 1e3, "per_layer_update_calls_us": t_lenet_upd_loop * 1e3,
                            "per_layer_update_calls_graph_us": t_lenet_upd_loop_graph * 1e3,
                            "layer_streams": {"per_layer_calls_us": t_lenet_forked * 1e3, "per_layer_calls_graph_us": t_lenet_forked_graph * 1e3,
                                              "per_layer_update_calls_us": t_lenet_upd_forked * 1e3,
                                              "per_layer_update_calls_graph_us": t_lenet_upd_forked_graph * 1e3,
                                              "note": "the same per-layer calls inside `with psgd_tf_amd.kron.layer_streams():` -- every "
                                                      "call on its own forked stream, joined when the block ends; `*_graph_us`: captured once, "
                                                      "the graph has one branch per layer"},
                            "batched_update_us": t_lenet_upd * 1e3},
        "4096x4096_fp32_update": {"ms": t_upd, "gflops": f_upd / t_upd / 1e6, "arithmetic": "f16x2-plane fp32 emulation (fp32 accumulation)",
                                  "ms_exact_fp32_mfma": t_upd_exact},
        "4096x4096_bf16_operands_update": {"ms": t_upd_bf16, "gflops": f_upd / t_upd_bf16 / 1e6,
                                           "note": "products on bf16 operands; balance, triangular solves, norms and "
                                                   "the final subtraction in fp32"},
    }


def splu_bench(dev, psgd, N=50_000_000, r=10, iters=10):
    """Secondary leg: sparse-LU preconditioner update + apply (psgd.py:396-524) at the rank the
    reference's demo uses (demo_usage_of_all_preconditioners.py:45).  Bytes are what the sweeps move."""
    g = torch.Generator(device=dev).manual_seed(3)
    n2, sc = N - r, 0.3 / r ** 0.5
    L12 = torch.randn(N, r, device=dev, generator=g) * (sc * 3 * (r / N) ** 0.5)
    U12 = torch.randn(r, N, device=dev, generator=g) * (sc * 3 * (r / N) ** 0.5)
    L12[:r] = torch.tril(torch.randn(r, r, device=dev, generator=g) * sc, -1) + torch.eye(r, device=dev)
    U12[:, :r] = torch.triu(torch.randn(r, r, device=dev, generator=g) * sc, 1) + torch.eye(r, device=dev)
    l3 = torch.exp(torch.empty(n2, 1, device=dev).uniform_(-0.5, 0.5, generator=g))
    u3 = torch.exp(torch.empty(n2, 1, device=dev).uniform_(-0.5, 0.5, generator=g)) * 0.7
    dx = torch.randn(N, 1, device=dev, generator=g)
    dg = dx * torch.exp(torch.empty(N, 1, device=dev).uniform_(-2.3, 2.3, generator=g))
    gr = torch.randn(N, 1, device=dev, generator=g)
    st = [L12, l3, U12, u3]

    def step():
        st[:] = psgd.update_precond_splu(*st, [dx], [dg], 0.01)
        return psgd.precond_grad_splu(*st, [gr])

    for _ in range(2):
        step()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    bytes_row = 4 * (9 * r + 15) + 4 * (3 * r + 9)     # update: 4 sweeps; apply: 3 sweeps
    return {"metric": "splu_update_apply_params_per_sec", "value": N / ms * 1e3, "unit": "params/s", "N": N, "r": r,
            "ms_per_step": ms, "bytes_per_param": bytes_row, "achieved_GBs": bytes_row * N / ms / 1e6,
            "frac_of_hbm_peak": bytes_row * N / ms / 1e6 / HBM_PEAK_GBS}


def wide_rank_bench(dev, psgd, N=20_000_000, iters=6):
    """Secondary leg (review item 8): ranks above 32 of both low-rank preconditioners beside a specialised rank at the same N --
    UVd r = 32 / 64 and sparse LU r = 32 / 40, apply and update separately, on the bytes the reference's arithmetic needs
    (UVd 4(4r+5) / 4(5r+10), sparse LU 4(3r+9) / 4(9r+15) per row: DESIGN 4.1, 4.5).  `x_spec` = rate / rate of the rank-32 line."""
    def timeit(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / iters

    out, base = {"N": N}, {}
    for r in (32, 64):
        g = torch.Generator(device=dev).manual_seed(r)
        sc = (1.0 / (N * r)) ** 0.5
        U, V = torch.randn(N, r, device=dev, generator=g) * sc, torch.randn(N, r, device=dev, generator=g) * sc
        d = torch.ones(N, 1, device=dev)
        gr, v = torch.randn(N, 1, device=dev, generator=g), torch.randn(N, 1, device=dev, generator=g)
        h = v * 1.5
        ta = timeit(lambda: psgd.precond_grad_UVd_math(U, V, d, gr))
        flip = [0]

        def upd():
            flip[0] ^= 1
            psgd.update_precond_UVd_math_(U, V, d, v, h, STEP, TINY, balance=False, update_U=bool(flip[0]))
        tu = timeit(upd)

        def fused():                                   # the UVd.step pattern: update, then apply on the updated state, one call
            flip[0] ^= 1
            return psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, gr, STEP, TINY, balance=False, update_U=bool(flip[0]))
        ts = timeit(fused)
        ra, ru, rs = 4 * (4 * r + 5) * N / ta / 1e6, 4 * (5 * r + 10) * N / tu / 1e6, 4 * (9 * r + 15) * N / ts / 1e6
        if r == 32:
            base["uvd"] = (ra, ru, rs)
        out["uvd_r%d" % r] = {"apply_ms": ta, "update_ms": tu, "step_ms": ts, "apply_GBs": ra, "update_GBs": ru, "step_GBs": rs,
                              "apply_x_spec": ra / base["uvd"][0], "update_x_spec": ru / base["uvd"][1],
                              "step_x_spec": rs / base["uvd"][2]}
        del U, V, d, gr, v, h
        torch.cuda.empty_cache()
    for r in (32, 40):
        g = torch.Generator(device=dev).manual_seed(3)
        n2, sc = N - r, 0.3 / r ** 0.5
        L12 = torch.randn(N, r, device=dev, generator=g) * (sc * 3 * (r / N) ** 0.5)
        U12 = torch.randn(r, N, device=dev, generator=g) * (sc * 3 * (r / N) ** 0.5)
        L12[:r] = torch.tril(torch.randn(r, r, device=dev, generator=g) * sc, -1) + torch.eye(r, device=dev)
        U12[:, :r] = torch.triu(torch.randn(r, r, device=dev, generator=g) * sc, 1) + torch.eye(r, device=dev)
        l3 = torch.exp(torch.empty(n2, 1, device=dev).uniform_(-0.5, 0.5, generator=g))
        u3 = torch.exp(torch.empty(n2, 1, device=dev).uniform_(-0.5, 0.5, generator=g)) * 0.7
        dx = torch.randn(N, 1, device=dev, generator=g)
        dg = dx * torch.exp(torch.empty(N, 1, device=dev).uniform_(-2.3, 2.3, generator=g))
        gr = torch.randn(N, 1, device=dev, generator=g)
        tu = timeit(lambda: psgd.update_precond_splu(L12, l3, U12, u3, [dx], [dg], 0.01))
        ta = timeit(lambda: psgd.precond_grad_splu(L12, l3, U12, u3, [gr]))
        ra, ru = 4 * (3 * r + 9) * N / ta / 1e6, 4 * (9 * r + 15) * N / tu / 1e6
        if r == 32:
            base["lu"] = (ra, ru)
        out["splu_r%d" % r] = {"apply_ms": ta, "update_ms": tu, "apply_GBs": ra, "update_GBs": ru,
                               "apply_x_spec": ra / base["lu"][0], "update_x_spec": ru / base["lu"][1]}
        del L12, U12, l3, u3, dx, dg, gr
        torch.cuda.empty_cache()
    return out


PROF_SLOTS = (("apply_s1", 0), ("apply_s2", 1), ("apply_s3", 2), ("update_s1", 3), ("update_s2", 4), ("update_s3", 5))


def prof_collect(lib):
    """Average duration (ms) per launch of every sweep kernel since psgd_prof_enable(1)."""
    out = {}
    for name, slot in PROF_SLOTS:
        tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
        lib.psgd_prof_collect(slot, ctypes.byref(tot), ctypes.byref(cnt))
        out[name] = (tot.value / cnt.value) if cnt.value else None
    return out


def uvd_bytes(r, fused):
    """Bytes per row per launch that each sweep kernel moves."""
    kb = {"apply_s1": 4 * (r + 2), "apply_s2": 4 * (r + 3), "apply_s3": 4 * (r + 3),
          "update_s1": 4 * (2 * r + 3), "update_s2": 4 * (3 * r + 4), "update_s3": 12}
    if fused:
        kb["update_s2"] = 4 * (3 * r + 5)      # + 4 B/row for g; also reduces [Unew | Vnew]' [d.*g, d.*g.*nablaD]
        kb["apply_s3"] = 4 * (2 * r + 5)       # the fused step's last sweep: d update + whole apply (k_uvd_final)
        del kb["apply_s1"], kb["apply_s2"], kb["update_s3"]
    return kb


def uvd_legs(dev, psgd, lib, state, r, iters, arena=None):
    """precond_grad_UVd_math alone, update_precond_UVd_math_ alone (the two reference-named calls) and the fused
    step on one resident problem: wall time per call from HIP events on the launch stream, per-kernel times from
    the psgd_prof hooks, HBM fractions on SURVEY 8d's algorithmic bytes and on the bytes the sweeps move."""
    U, V, d, grad, v, h = state
    n = U.shape[0]

    def timed(fn, count):
        for i in range(2):
            fn(i)
        torch.cuda.synchronize(dev)
        lib.psgd_prof_enable(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(count):
            fn(i)
        e1.record()
        torch.cuda.synchronize(dev)
        wall = e0.elapsed_time(e1) / count
        lib.psgd_prof_enable(1)                 # second pass: same calls with the per-kernel event hooks on
        for i in range(count):
            fn(i)
        torch.cuda.synchronize(dev)
        k = prof_collect(lib)
        lib.psgd_prof_enable(0)
        return wall, k

    def leg(wall, kms, names, alg, moved):
        ksum = sum(kms[x] or 0.0 for x in names)
        gbs = lambda b, ms: b * n / (ms * 1e-3) / 1e9
        return {"wall_ms": wall, "kernel_ms": ksum, "params_per_s": n / (wall * 1e-3),
                "alg_bytes_per_param": alg, "moved_bytes_per_param": moved,
                "frac": gbs(alg, wall) / HBM_PEAK_GBS, "frac_moved": gbs(moved, wall) / HBM_PEAK_GBS,
                "frac_kernels_only": gbs(alg, ksum) / HBM_PEAK_GBS if ksum else None,
                "kernels_ms": {x: kms[x] for x in names}}

    okw = {"out": arena.out} if arena is not None else {}          # (the state's owner also owns where the result is written)
    wa, ka = timed(lambda i: psgd.precond_grad_UVd_math(U, V, d, grad, **okw), iters)
    wu, ku = timed(lambda i: psgd.update_precond_UVd_math_(U, V, d, v, h, STEP, TINY, balance=False,
                                                           update_U=(i % 2 == 0)), iters)
    wf, kf = timed(lambda i: psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, grad, STEP, TINY, balance=False,
                                                                           update_U=(i % 2 == 0), **okw), iters)
    return {
        "N": n, "r": r, "timed_calls": iters,
        "apply": dict(leg(wa, ka, ("apply_s1", "apply_s2", "apply_s3"), 4 * (4 * r + 5), 4 * (3 * r + 8)),
                      call="precond_grad_UVd_math (psgd.py:619-627)"),
        "update": dict(leg(wu, ku, ("update_s1", "update_s2", "update_s3"), 4 * (5 * r + 10), 4 * (5 * r + 10)),
                       call="update_precond_UVd_math_ (psgd.py:554-617)"),
        "step_fused": dict(leg(wf, kf, ("update_s1", "update_s2", "apply_s3"),
                               4 * (9 * r + 15), 4 * (2 * r + 3) + 4 * (3 * r + 5) + 4 * (2 * r + 5)),
                           call="update_precond_UVd_math_and_precond_grad (psgd.py:732 -> :748)"),
    }


_DROP_KEYS = {"note", "call", "timing", "batched_call", "flop_count", "runs_ms", "mfma_pmc", "sample_detail", "kernels_ms_detail", "log", "search"}


def _no_nan(x):
    """NaN / inf are not JSON: a failed leg's placeholder becomes null"""
    if isinstance(x, dict):
        return {k: _no_nan(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_no_nan(v) for v in x]
    if isinstance(x, float) and (x != x or x in (float("inf"), float("-inf"))):
        return None
    return x


def _strip(x):
    """The full record without its prose: drops the keys of _DROP_KEYS at any depth and cuts strings to 120 characters (what
    the driver's record keeps of a string)."""
    if isinstance(x, dict):
        return {k: _strip(v) for k, v in x.items() if k not in _DROP_KEYS}
    if isinstance(x, list):
        return [_strip(v) for v in x]
    if isinstance(x, str) and len(x) > 120:
        return x[:117] + "..."
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.8g" % x)                 # (value / ms_per_step pairs stay consistent to 1e-7)
    return x


def compact_line(res, limit=7800):
    """The ONE line of the contract: every number of the full record, no prose (the key glossary is DESIGN.md section 5), with
    the figures the north star names hoisted into `roofline` as scalars -- apply_frac / apply_ms (precond_grad_UVd_math alone on
    SURVEY's 340 B/param), update_frac, step_two_reference_calls_ms, kron_bf16_apply_frac_issued, kron_mfma_util ... -- because the
    driver's record keeps the scalar entries of `roofline` and `cpu_baseline` and an 8-KB tail."""
    line = _strip(res)
    for k in ("value", "ms_per_step"):           # the contract's own numbers: untouched
        line[k] = res[k]
    rf = line["roofline"]
    paths = rf.get("paths", {})
    hoist = {}
    if "apply" in paths:
        a, u = paths["apply"], paths["update"]
        hoist.update(apply_ms=a["wall_ms"], apply_frac=a["frac"], apply_frac_moved=a["frac_moved"],
                     update_ms=u["wall_ms"], update_frac=u["frac"],
                     step_two_reference_calls_ms=paths["step_two_reference_calls"]["wall_ms"])
    if "step" in paths:
        hoist.update(step_frac=paths["step"]["frac"], step_frac_moved=paths["step"]["frac_moved"])
    kr = res.get("kron")
    if kr:
        hoist.update(kron_bf16_apply_ms=kr["4096x4096_bf16_operands"]["ms"],
                     kron_bf16_apply_gflops_Fref=kr["4096x4096_bf16_operands"]["gflops"],
                     kron_bf16_apply_frac_issued=kr["roofline"]["frac"],
                     kron_fp32_apply_ms=kr["4096x4096_fp32"]["ms"], kron_fp32_update_ms=kr["4096x4096_fp32_update"]["ms"],
                     kron_bf16ops_update_ms=kr["4096x4096_bf16_operands_update"]["ms"],
                     kron_fp32_apply_auto_route_ms=kr["4096x4096_fp32"]["ms_auto_route"],
                     kron_fp32_exact_apply_ms=kr["4096x4096_fp32"]["ms_exact_fp32_mfma"],
                     kron_fp32_exact_update_ms=kr["4096x4096_fp32_update"]["ms_exact_fp32_mfma"],
                     lenet5_apply_us=kr["lenet5_set_fp32"]["us"], lenet5_update_us=kr["lenet5_set_fp32"]["update_us"],
                     lenet5_layer_batch_apply_us=kr["lenet5_set_fp32"]["layer_batch_us"],
                     lenet5_layer_batch_update_us=kr["lenet5_set_fp32"]["layer_batch_update_us"],
                     lenet5_batched_apply_us=kr["lenet5_set_fp32"]["batched_us"],
                     lenet5_batched_update_us=kr["lenet5_set_fp32"]["batched_update_us"])
        pm = kr["4096x4096_bf16_operands"].get("mfma_pmc")
        if pm:
            hoist["kron_mfma_util"] = pm["MfmaUtil_percent_time_weighted"] / 100.0
    c2 = res.get("config2_N1M_r10")
    if c2:
        hoist["config2_step_us"] = c2["step_fused"]["wall_ms"] * 1e3
    wr = res.get("wide_rank")
    if wr:
        hoist.update(uvd_r64_apply_ms=wr["uvd_r64"]["apply_ms"], uvd_r64_update_ms=wr["uvd_r64"]["update_ms"],
                     uvd_r64_apply_x_spec=wr["uvd_r64"]["apply_x_spec"], uvd_r64_update_x_spec=wr["uvd_r64"]["update_x_spec"],
                     uvd_r64_step_ms=wr["uvd_r64"]["step_ms"], uvd_r64_step_x_spec=wr["uvd_r64"]["step_x_spec"],
                     splu_r40_apply_x_spec=wr["splu_r40"]["apply_x_spec"], splu_r40_update_x_spec=wr["splu_r40"]["update_x_spec"])
    rf.update({k: float("%.6g" % v) for k, v in hoist.items() if v is not None})
    # size guard: drop the least important sub-records first (they stay in the BENCH_DETAIL line on stderr)
    for victim in (("kron", "lenet5_set_fp32", "layer_streams"), ("roofline", "paths", "step_fused_events"),
                   ("config2_N1M_r10", "step_fused", "kernels_ms"), ("config2_N1M_r10", "apply", "kernels_ms"),
                   ("config2_N1M_r10", "update", "kernels_ms"), ("wide_rank",), ("splu",), ("config2_N1M_r10",), ("roofline", "kernels")):
        if len(json.dumps(line)) <= limit:
            break
        node = line
        for k in victim[:-1]:
            node = node.get(k, {}) if isinstance(node, dict) else {}
        if isinstance(node, dict):
            node.pop(victim[-1], None)
    return line


def launch_ranks(args):
    """`python bench.py --gpus N` from a plain shell: start the N ranks as a child torch.distributed.run job.  This
    parent never initialises a GPU (device_count() does not), so nothing that holds the device is ever replaced."""
    import socket
    import subprocess
    single_dev = os.environ.get("PSGD_BENCH_SINGLE_DEVICE", "0") == "1"
    have = torch.cuda.device_count()
    if not single_dev and have < args.gpus:
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, have), file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if proc.returncode != 0 else (0 if lines else 1)


def run_uvd(args, psgd, sharded, lib, dev, rank, world, use_dist, n_local, n_global, steps, warmup, keep_state=False,
            placement_mode=None):
    """Warm-up + timed steps of the UVd update+apply on this rank's rows: `steps` steps between barrier + synchronize
    (MAX over ranks), then a second pass of the same steps with the per-kernel HIP event hooks on."""
    import torch.distributed as dist
    r = args.rank_r
    # the state's owner places it (psgd_tf_amd/placement.py): one allocation for U, V, d, the workspace and the output, the layout
    # chosen by a timed probe of both branches (results are bit-identical: only addresses change).
    arena, placement_log = None, []
    mode = placement_mode or args.placement
    if mode != "none" and r <= 32:                          # (per rank: the probe is local, no collective in it)
        from psgd_tf_amd import placement
        arena = (placement.UVdArena.probe(n_local, r, dev, log=placement_log) if mode == "probe"
                 else placement.UVdArena.packed(n_local, r, dev))
        arena.install_workspace()
    U, V, d, grad, v, h = make_inputs(n_local, n_global, r, dev, seed=1000 * rank, arena=arena)
    out_kw = {"out": arena.out} if arena is not None else {}
    if arena is not None and mode == "probe" and args.settle_s > 0:
        # the probe has just handed tens of GiB back to the driver, which clears released VRAM in the background (on the copy engines, at
        # the expense of HBM bandwidth): let that finish before anything is timed (untimed set-up, like the probe itself)
        torch.cuda.synchronize(dev)
        time.sleep(args.settle_s)
    # one step = update then apply on the updated state (psgd.py:732 -> :748).  Default: the fused call
    # (identical results, one pass over V less); --unfused times the two reference-named calls back to back.
    mod = sharded if use_dist else psgd
    if args.unfused:
        def step(i):
            mod.update_precond_UVd_math_(U, V, d, v, h, STEP, TINY, balance=False, update_U=(i % 2 == 0))
            return mod.precond_grad_UVd_math(U, V, d, grad)
    else:
        def step(i):
            return mod.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, grad, STEP, TINY, balance=False,
                                                                update_U=(i % 2 == 0), **out_kw)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- headline pass: no profiling hooks inside the timed region
    lib.psgd_prof_enable(0)
    for i in range(warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(out).all().item(), "non-finite preconditioned gradient"
    # ---- second pass: the same steps with a HIP event pair around every sweep launch (per-kernel durations)
    lib.psgd_prof_enable(1)
    for i in range(steps):
        step(i)
    fence()
    slot_ms = prof_collect(lib)
    lib.psgd_prof_enable(0)
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    rec = {"elapsed": elapsed, "slot_ms": slot_ms, "n_local": n_local, "n_global": n_global,
           "ms_per_step": elapsed / steps * 1e3, "value": n_global * steps / elapsed}
    if arena is not None:
        inf = arena.info
        rec["placement"] = {"mode": mode, "layout": inf.get("layout"), "held_gib": arena.bytes_held / 2**30,
                            "note": inf.get("note"), "probe_packed_step_ms": inf.get("packed_step_ms"),
                            "probe_step_ms": None if "step_U_ms" not in inf else 0.5 * (inf["step_U_ms"] + inf["step_V_ms"]),
                            "candidates": inf.get("candidates"), "search": inf.get("search"), "log": placement_log}
    if keep_state:
        rec["state"] = (U, V, d, grad, v, h)
        rec["arena"] = arena
    return rec


def exchange_overhead(args, psgd, sharded, lib, dev, rows, steps):
    """What the multi-GPU path adds to a step on ONE rank's share of BASELINE configs[3] (12.5M rows, r = 20):
    the fused step run unsharded, then through psgd_tf_amd/sharded.py on a 1-rank RCCL group (same sweeps + 2
    all-gathers + 2 fold kernels + the r x r kernels as separate stage calls).  The xGMI latency of a W-rank
    all-gather is not in it (one GPU here); host issue, the collectives' launches and the folds are."""
    import torch.distributed as dist
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
    try:
        import statistics
        # ONE set of tensors for both paths (round 6): where d / out / the workspace land relative to the factors is worth +-2 % of a step
        #  -- more than the exchanges cost -- so two separately allocated problems cannot be compared at this resolution
        r = args.rank_r
        U, V, d, grad, v, h = make_inputs(rows, rows, r, dev, seed=0)
        paths = {"unsharded": psgd, "sharded_1rank": sharded}

        def timed(mod, count):
            for i in range(10):
                mod.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, grad, STEP, TINY, balance=False, update_U=(i % 2 == 0))
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(count):
                out = mod.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, grad, STEP, TINY, balance=False, update_U=(i % 2 == 0))
            torch.cuda.synchronize(dev)
            assert torch.isfinite(out).all().item()
            return (time.perf_counter() - t0) / count * 1e3
        runs = {"unsharded": [], "sharded_1rank": []}
        for rnd in range(5):                                 # interleaved rounds: the two paths see the same clocks
            for name in ("unsharded", "sharded_1rank"):
                runs[name].append(timed(paths[name], steps))
        base, shd = statistics.median(runs["unsharded"]), statistics.median(runs["sharded_1rank"])
        return {"rows": rows, "r": r, "steps": steps, "rounds": 5, "unsharded_ms": base,
                "sharded_1rank_rccl_ms": shd, "added_us_per_step": (shd - base) * 1e3,
                "added_frac_of_step": (shd - base) / base,
                "backend": dist.get_backend(), "direct_rccl": sharded._direct_comm(None, dev) is not None, "runs_ms": runs,
                "note": "one rank's share of BASELINE configs[3] (100M rows / 8) on ONE set of tensors; 1-rank RCCL group on this GPU: host "
                        "issue + 2 all-gathers on the caller's stream (the library's own RCCL communicator) + 2 fold kernels "
                        "per step, no xGMI hop; medians of 5 interleaved rounds"}
    finally:
        if own_group:
            dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=None,
                    help="weak scaling: rows of the flat parameter vector PER GPU (global = rows x gpus)")
    ap.add_argument("--global-rows", type=int, default=None,
                    help="strong scaling: global rows, split over the ranks in contiguous blocks (default 100M = BASELINE configs[3])")
    ap.add_argument("--no-weak-leg", action="store_true", help="N > 1 default mode: skip the 100M-rows-per-GPU weak sub-record")
    ap.add_argument("--no-exchange-leg", action="store_true", help="N = 1: skip the exchange_overhead leg (1-rank RCCL group)")
    ap.add_argument("--rank-r", type=int, default=20, help="rank of modification r")
    ap.add_argument("--cpu-sample-rows", type=int, default=4_000_000)
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="fix the thread count of the CPU baseline (0 = sweep)")
    ap.add_argument("--cpu-full", action="store_true", help="(default since round 6; kept for old command lines)")
    ap.add_argument("--cpu-sample-only", action="store_true",
                    help="CPU baseline on the bounded sample only (default: the sample, then 1 + 2 steps on the metric's full N when the "
                         "host has >= 64 GiB available)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kron", action="store_true", help="skip the Kron apply GFLOP/s and sparse-LU legs")
    ap.add_argument("--no-legs", action="store_true", help="skip the apply-alone / update-alone / config-2 legs")
    ap.add_argument("--no-wide-rank", action="store_true", help="skip the rank 33..64 leg (UVd r = 64, sparse LU r = 40 beside r = 32)")
    ap.add_argument("--wide-rows", type=int, default=20_000_000, help="rows of the rank 33..64 leg")
    ap.add_argument("--unfused", action="store_true",
                    help="time update_precond_UVd_math_ + precond_grad_UVd_math as two separate calls")
    ap.add_argument("--bpc", type=int, default=0, help="experiment: cap on blocks per CU of the sweeps (psgd_set_tuning key 1)")
    ap.add_argument("--detail-json", default=None, help="also write the full record (every leg, notes) to this file; stdout "
                    "carries the compact line (< 8 KB), stderr a BENCH_DETAIL line with the full record")
    ap.add_argument("--placement", default="probe", choices=("probe", "packed", "none"),
                    help="unsharded run: who owns U, V, d, the workspace and the output -- 'probe' (default): one allocation, layout "
                         "chosen by a timed probe (psgd_tf_amd/placement.py); 'packed': one exact-size allocation; 'none': separate "
                         "torch allocations (rounds 1-5)")
    ap.add_argument("--settle-s", type=float, default=1.0,
                    help="seconds to wait after the placement probe has freed its search buffers (the driver clears released VRAM in the "
                         "background), before the warm-up steps; untimed set-up")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the multi-GPU code path (process group + exchanges) even at world size 1")
    args = ap.parse_args()
    if args.rows is not None and args.global_rows is not None:
        raise SystemExit("--rows (weak) and --global-rows (strong) are exclusive")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))            # before anything touches a GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # Test hook (not a measurement): PSGD_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 with gloo collectives, so the
    # N > 1 code path of this script can be exercised where only one GPU exists (RCCL refuses two ranks on one device).
    single_dev = os.environ.get("PSGD_BENCH_SINGLE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if single_dev else local_rank)
    torch.cuda.set_device(dev)

    import torch.distributed as dist
    use_dist = world > 1 or args.force_sharded
    backend = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if single_dev:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        backend = dist.get_backend()

    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import _lib, sharded
    lib = _lib.load()            # fails loudly if the HIP extension is missing
    if args.bpc:
        lib.psgd_set_tuning(1, args.bpc)

    r = args.rank_r
    # ---- which rows (see the module docstring)
    weak_leg_rows = None
    if args.rows is not None:
        scaling, n_local, n_global = "weak", args.rows, args.rows * world
    else:
        n_global = args.global_rows if args.global_rows is not None else 100_000_000
        lo, hi = sharded.shard_rows(n_global, rank, world)
        scaling, n_local = ("strong" if world > 1 or args.global_rows is not None else "weak"), hi - lo
        if args.global_rows is None and world > 1 and not args.no_weak_leg:
            weak_leg_rows = 100_000_000
    if n_local < 1:
        raise SystemExit("rank %d would hold no rows (global %d over %d ranks)" % (rank, n_global, world))

    main_rec = run_uvd(args, psgd, sharded, lib, dev, rank, world, use_dist, n_local, n_global, args.steps, args.warmup,
                       keep_state=(world == 1 and not args.no_legs))
    state = main_rec.pop("state", None)
    arena_keep = main_rec.pop("arena", None)      # (the legs below run on the same placed state)
    weak_rec = None
    if weak_leg_rows:
        torch.cuda.empty_cache()
        weak_rec = run_uvd(args, psgd, sharded, lib, dev, rank, world, use_dist, weak_leg_rows, weak_leg_rows * world,
                           args.steps, args.warmup)

    if rank == 0:
        slot_ms, ms_per_step, value = main_rec["slot_ms"], main_rec["ms_per_step"], main_rec["value"]
        kbytes = uvd_bytes(r, not args.unfused)
        kern = {k: {"avg_ms": slot_ms[k], "achieved_GBs": kbytes[k] * n_local / (slot_ms[k] * 1e-3) / 1e9}
                for k in kbytes if slot_ms[k]}
        dom = "update_s2"
        traffic, traffic_source = None, None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                rec = json.load(open(tp)).get("k_update_s2")
                if rec and rec.get("rows") == n_local and rec.get("r") == r:
                    traffic = rec["hbm_bytes_per_launch"]
                    traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc passes on an earlier box, NOT counters of this run)"
            except Exception:
                traffic = None
        ach = kern[dom]["achieved_GBs"]
        alg_step = 4 * (9 * r + 15)
        moved_step = (4 * (5 * r + 10) + 4 * (3 * r + 8)) if args.unfused else \
            (4 * (2 * r + 3) + 4 * (3 * r + 5) + 4 * (2 * r + 5))
        gbs = lambda b: b * n_local / (ms_per_step * 1e-3) / 1e9
        paths = {"fused": not args.unfused,
                 "step": {"alg_bytes_per_param": alg_step, "moved_bytes_per_param": moved_step, "wall_ms": ms_per_step,
                          "frac": gbs(alg_step) / HBM_PEAK_GBS, "frac_moved": gbs(moved_step) / HBM_PEAK_GBS,
                          "kernel_ms": sum(slot_ms[k] or 0.0 for k in slot_ms)}}
        if scaling == "strong":
            what = "N=%d rows global over %d GPU(s), contiguous blocks (%d on rank 0)" % (n_global, world, n_local)
        else:
            what = "N=%d rows per GPU" % n_local
        res = {
            "metric": "uvd_update_apply_params_per_sec", "value": value, "unit": "params/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": (scaling if world > 1 else None), "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "UVd update+apply (psgd.py:732->:748%s), %s, r=%d" % ("" if args.unfused else ", fused call", what, r),
                       "rows_per_gpu": n_local, "rows_global": n_global, "rank_of_modification": r,
                       "baseline_config": ("configs[3]: UVd N=100M, r=20, flat-vector sharded" if
                                           (scaling == "strong" and n_global == 100_000_000 and r == 20 and world > 1) else
                                           ("metric config: UVd N=100M, r=20, 1 GPU" if
                                            (n_global == 100_000_000 and r == 20 and world == 1) else None)),
                       "parallelism": ("row-sharded x%d: %d all-gathers/step of r-dim buffers (<= 30 KB) + rank-order fold"
                                       % (world, 4 if args.unfused else 2) if use_dist else "one GPU, no exchange") +
                                      (" [TEST MODE: all ranks on one GPU, gloo -- not a measurement]" if single_dev else ""),
                       "collective_backend": backend, "rccl_ranks": (dist.get_world_size() if use_dist else None),
                       "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version()) if (use_dist and backend == "nccl") else None),
                       "step": STEP, "branches": "balance=0, update_U alternating",
                       "placement": main_rec.get("placement")},
            "roofline": {"bound": "hbm", "kernel": "k_update_s2 (update sweep 2, dominant kernel)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "alg_bytes_per_launch": kbytes[dom] * n_local,
                         "avg_launch_ms": slot_ms[dom], "kernels": kern, "paths": paths,
                         "timing": "headline pass without profiling hooks; kernel durations from a second pass of the "
                                   "same steps with HIP event pairs on the launch stream"},
        }
        if weak_rec is not None:
            wk = weak_rec["slot_ms"]
            res["weak"] = {"scaling": "weak", "value": weak_rec["value"], "unit": "params/s",
                           "ms_per_step": weak_rec["ms_per_step"], "rows_per_gpu": weak_rec["n_local"],
                           "rows_global": weak_rec["n_global"], "steps": args.steps, "warmup": args.warmup,
                           "frac_of_hbm_roofline_per_gpu": alg_step * weak_rec["n_local"] /
                           (weak_rec["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "kernels_ms": {k: wk[k] for k in kbytes if wk[k]}}
        if world == 1:
            if state is not None:
                legs = uvd_legs(dev, psgd, lib, state, r, max(5, min(args.steps, 20)), arena=arena_keep)
                paths["apply"], paths["update"], paths["step_fused_events"] = legs["apply"], legs["update"], legs["step_fused"]
                # the two reference-named calls back to back, next to the fused entry point the headline uses
                two = legs["update"]["wall_ms"] + legs["apply"]["wall_ms"]
                paths["step_two_reference_calls"] = {
                    "wall_ms": two, "params_per_s": n_local / (two * 1e-3),
                    "frac": alg_step * n_local / (two * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "call": "update_precond_UVd_math_ then precond_grad_UVd_math (psgd.py:732, :748), separately timed"}
            state = arena_keep = None
            torch.cuda.empty_cache()
            if not args.no_legs:
                c2 = make_inputs(1_000_000, 1_000_000, 10, dev, seed=7)
                legs2 = uvd_legs(dev, psgd, lib, c2, 10, 200)
                legs2["note"] = ("BASELINE config 2: 88 MB working set, resident in the 256 MiB Infinity Cache -> "
                                 "launch/latency-bound; fractions are against the HBM roof for reference only")
                res["config2_N1M_r10"] = legs2
                del c2
                torch.cuda.empty_cache()
            if not args.no_kron:
                res["kron"] = kron_bench(dev, psgd)
                res["splu"] = splu_bench(dev, psgd)
            if not args.no_wide_rank:
                res["wide_rank"] = wide_rank_bench(dev, psgd, N=args.wide_rows)
            # (after the Kron leg: a process that has initialised an RCCL communicator runs the two-stream Kron updates
            # 1.1-1.3 ms slower at 4096^2 -- tools/rccl_fork_probe.py, profiles/r03_rccl_fork_probe.txt)
            if not args.no_exchange_leg and not single_dev and not args.force_sharded:
                try:
                    res["exchange_overhead"] = exchange_overhead(args, psgd, sharded, lib, dev,
                                                                 sharded.shard_rows(100_000_000, 0, 8)[1], 100)
                except Exception as e:                      # a leg, never the headline: report, do not fail the line
                    res["exchange_overhead"] = {"error": "%s: %s" % (type(e).__name__, e)}
                torch.cuda.empty_cache()
            if not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(r, args.cpu_sample_rows, args.cpu_budget_s, args.cpu_threads or None,
                                                   full_rows=None if args.cpu_sample_only else n_local)
        res = _no_nan(res)
        detail = json.dumps(res)
        if args.detail_json:
            with open(args.detail_json, "w") as fh:
                fh.write(detail + "\n")
        print("BENCH_DETAIL " + detail, file=sys.stderr, flush=True)
        try:                                     # RCCL prints its version banner through C stdio, which a pipe buffers until exit:
            ctypes.CDLL(None).fflush(None)       # out with it now, so that the JSON line is the LAST line of stdout
        except Exception:
            pass
        print(json.dumps(compact_line(res)), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
