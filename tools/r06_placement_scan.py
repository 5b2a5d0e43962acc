"""Round 6, review item 1: how the fused UVd step's time depends on WHERE its streams live.

All eight regions of a step -- U, V, d, g, v, h, out and the workspace (which holds nablaD) -- are carved out of ONE device
allocation at offsets this script chooses, the fused call of the headline (psgd_uvd_update_apply_f32) is timed on both branches
with the per-kernel HIP-event hooks on, and one JSON line per layout goes to --out.  Layouts: the bench's own separate
allocations, the packed slab, one gap at a time over a ladder of sizes, region orders, and random layouts.

  python tools/r06_placement_scan.py --out gpurun_out/r06_placement_scan.jsonl [--random 150]
"""
import argparse
import ctypes
import itertools
import json
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from psgd_tf_amd import _lib  # noqa: E402

NAMES = ("U", "V", "d", "g", "v", "h", "out", "ws")
SLOTS = (("update_s1", 3), ("update_s2", 4), ("final", 2))


def region_bytes(N, r, ws_bytes):
    return {"U": 4 * N * r, "V": 4 * N * r, "d": 4 * N, "g": 4 * N, "v": 4 * N, "h": 4 * N, "out": 4 * N, "ws": ws_bytes}


def place(order, gaps, sizes, align=256):
    """offsets of the regions laid out in `order`, `gaps[name]` bytes of padding BEFORE region name"""
    off, cur = {}, 0
    for name in order:
        cur += gaps.get(name, 0)
        cur = -(-cur // align) * align
        off[name] = cur
        cur += sizes[name]
    return off, cur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=100_000_000)
    ap.add_argument("--r", type=int, default=20)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--random", type=int, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/r06_placement_scan.jsonl")
    ap.add_argument("--slack-mb", type=int, default=1536)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--mode", default="layouts", choices=("layouts", "scan1d"))
    ap.add_argument("--slab-gib", type=float, default=0.0, help="scan1d: size of the slab the regions slide in")
    args = ap.parse_args()
    N, r = args.N, args.r
    dev = torch.device("cuda:0")
    lib = _lib.load()
    ws_bytes = int(lib.psgd_uvd_workspace_bytes(N, r))
    sizes = region_bytes(N, r, ws_bytes)
    total = sum(sizes.values())
    slab_bytes = int(args.slab_gib * 2**30) if args.mode == "scan1d" else total + args.slack_mb * 2**20
    slab = torch.empty(slab_bytes, dtype=torch.uint8, device=dev)
    base = slab.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    scale = (1.0 / (N * r)) ** 0.5
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    fout = open(args.out, "a")

    def views(off):
        t = {}
        for k in NAMES:
            raw = slab[off[k]:off[k] + sizes[k]]
            t[k] = raw if k == "ws" else raw.view(torch.float32)
        return t

    def fill(t):
        g = torch.Generator(device=dev).manual_seed(0)
        t["U"].normal_(generator=g).mul_(scale)
        t["V"].normal_(generator=g).mul_(scale)
        t["d"].fill_(1.0)
        t["g"].normal_(generator=g)
        t["v"].normal_(generator=g)
        t["h"].uniform_(-4.6, 4.6, generator=g).exp_().mul_(t["v"])

    def timed(ptr, label, extra):
        def call(update_u):
            rc = lib.psgd_uvd_update_apply_f32(ptr["U"], ptr["V"], ptr["d"], ptr["v"], ptr["h"], ptr["g"], ptr["out"], N, r,
                                               0.01, 1.1754943508222875e-38, 0, update_u, ptr["ws"], ws_bytes, st)
            assert rc == 0, rc
        rec = {"label": label}
        rec.update(extra)
        lib.psgd_prof_enable(0)
        call(1)
        call(0)
        for br, name in ((1, "U"), (0, "V")):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                call(br)
            e1.record()
            torch.cuda.synchronize()
            rec["step_%s_ms" % name] = e0.elapsed_time(e1) / args.iters
            lib.psgd_prof_enable(1)
            for _ in range(args.iters):
                call(br)
            torch.cuda.synchronize()
            for kname, slot in SLOTS:
                tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
                lib.psgd_prof_collect(slot, ctypes.byref(tot), ctypes.byref(cnt))
                rec["%s_%s_ms" % (kname, name)] = tot.value / max(cnt.value, 1)
            lib.psgd_prof_enable(0)
        rec["step_alt_ms"] = 0.5 * (rec["step_U_ms"] + rec["step_V_ms"])
        fout.write(json.dumps(rec) + "\n")
        fout.flush()
        print("%-34s step U %.3f V %.3f | s1 %.3f %.3f | s2 %.3f %.3f | fin %.3f %.3f" % (
            label[:34], rec["step_U_ms"], rec["step_V_ms"], rec["update_s1_U_ms"], rec["update_s1_V_ms"],
            rec["update_s2_U_ms"], rec["update_s2_V_ms"], rec["final_U_ms"], rec["final_V_ms"]), flush=True)
        return rec

    def run_layout(label, order, gaps):
        off, end = place(order, gaps, sizes)
        assert end <= slab.numel(), (label, end, slab.numel())
        t = views(off)
        fill(t)
        ptr = {k: t[k].data_ptr() for k in NAMES}
        return timed(ptr, label, {"order": list(order), "gaps": {k: int(v) for k, v in gaps.items()},
                                  "off": {k: int(v) for k, v in off.items()}, "base": base})

    # 0. the bench's own layout: separate allocations in make_inputs' order
    def separate(label, swap=False):
        g = torch.Generator(device=dev).manual_seed(0)
        t = {}
        first, second = ("V", "U") if swap else ("U", "V")
        t[first] = torch.randn(N, r, device=dev, generator=g) * scale
        t[second] = torch.randn(N, r, device=dev, generator=g) * scale
        t["d"] = torch.ones(N, 1, device=dev)
        t["g"] = torch.randn(N, 1, device=dev, generator=g)
        t["v"] = torch.randn(N, 1, device=dev, generator=g)
        t["h"] = t["v"] * torch.exp(torch.empty(N, 1, device=dev).uniform_(-4.6, 4.6, generator=g))
        t["ws"] = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        t["out"] = torch.empty(N, 1, device=dev)
        ptr = {k: t[k].data_ptr() for k in NAMES}
        rec = timed(ptr, label, {"ptr": {k: int(v) for k, v in ptr.items()}})
        del t
        torch.cuda.empty_cache()
        return rec

    def run_offsets(label, off):
        for k in NAMES:
            assert off[k] % 256 == 0 and 0 <= off[k] and off[k] + sizes[k] <= slab.numel(), (label, k, off[k])
        spans = sorted((off[k], off[k] + sizes[k]) for k in NAMES)
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), (label, "overlap")
        t = views(off)
        fill(t)
        ptr = {k: t[k].data_ptr() for k in NAMES}
        return timed(ptr, label, {"off": {k: int(v) for k, v in off.items()}, "base": base})

    canon = ("U", "V", "d", "g", "v", "h", "out", "ws")
    t0 = time.time()
    if args.mode == "scan1d":
        G = 2**30
        q = lambda x: int(x) // 256 * 256
        thin = ("d", "g", "v", "h", "out", "ws")
        fsz, tsz = sizes["U"], sum(sizes[k] for k in thin) + 6 * 256

        def thin_at(start, order=thin):
            off, cur = {}, q(start)
            for k in order:
                off[k] = cur
                cur = q(cur + sizes[k] + 255)
            return off
        top = slab.numel()
        # D: the packed block slides as a whole (absolute position only)
        packed, _ = place(canon, {}, sizes)
        x = 0.0
        while q(x * G) + total + 8 * 256 <= top:
            run_offsets("D block@%.2f" % x, {k: packed[k] + q(x * G) for k in NAMES})
            x += 1.0
        # A: V at 0, thin streams right after it, U slides over the rest
        fixed = dict(thin_at(fsz + 256), V=0)
        x = (fsz + tsz) / G + 0.05
        while q(x * G) + fsz <= top:
            run_offsets("A U@%.2f" % x, dict(fixed, U=q(x * G)))
            x += 0.5
        # A2: U at 0, thin after it, V slides
        fixed = dict(thin_at(fsz + 256), U=0)
        x = (fsz + tsz) / G + 0.05
        while q(x * G) + fsz <= top:
            run_offsets("A2 V@%.2f" % x, dict(fixed, V=q(x * G)))
            x += 0.5
        # B: U at 0, V right after, the thin block slides
        x = 2 * fsz / G + 0.05
        while q(x * G) + tsz <= top:
            run_offsets("B thin@%.2f" % x, dict(thin_at(x * G), U=0, V=q(fsz + 256)))
            x += 0.5
        # C: as packed, only d / out / ws(nablaD) / g slides (one at a time) over the free space behind the block
        for name in ("d", "out", "ws", "g"):
            rest = [k for k in canon if k != name]
            base_off, end = place(rest, {}, sizes)
            x = end / G + 0.05
            while q(x * G) + sizes[name] <= top:
                run_offsets("C %s@%.2f" % (name, x), dict(base_off, **{name: q(x * G)}))
                x += 1.0
        print("scan done in %.1f s" % (time.time() - t0))
        return
    for rep in range(2):
        run_layout("packed#%d" % rep, canon, {})
    if not args.quick:
        ladder = [256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 2**20, 2**21, 2**22, 2**24, 2**26,
                  2**28]
        for name in ("V", "d", "ws", "g", "out"):
            for gap in ladder:
                run_layout("gap %s %d" % (name, gap), canon, {name: gap})
        orders = [("V", "U", "d", "g", "v", "h", "out", "ws"), ("d", "g", "v", "h", "out", "ws", "U", "V"),
                  ("U", "d", "g", "v", "h", "out", "ws", "V"), ("ws", "U", "V", "d", "g", "v", "h", "out"),
                  ("U", "ws", "V", "d", "g", "v", "h", "out"), ("d", "U", "g", "v", "V", "h", "out", "ws"),
                  ("U", "V", "ws", "out", "d", "g", "v", "h")]
        for o in orders:
            run_layout("order " + "".join(x[0] if x != "ws" else "w" for x in o), o, {})
    rng = random.Random(args.seed)
    budget = args.slack_mb * 2**20 - 8 * 256
    for i in range(args.random):
        o = list(canon)
        rng.shuffle(o)
        gaps, left = {}, budget
        for name in o:
            kind = rng.random()
            if kind < 0.35:
                gp = 0
            elif kind < 0.8:
                gp = 256 * rng.randrange(1, 2**rng.randrange(1, 15))       # up to 4 MiB, log-uniform in scale
            else:
                gp = 256 * rng.randrange(1, 2**20)                           # up to 256 MiB
            gp = min(gp, left // 256 * 256)
            left -= gp
            gaps[name] = gp
        run_layout("rand#%d" % i, o, gaps)
    run_layout("packed#end", canon, {})
    del slab
    torch.cuda.empty_cache()
    separate("separate (bench order)")
    separate("separate (V before U)", swap=True)
    print("scan done in %.1f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
