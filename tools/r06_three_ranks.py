"""Round 6, placement: the region walk shows THREE classes of memory of 96 GiB each (a third of the 4-GiB allocations behave like the
first one, two thirds do not): the three ranks of the 12-high HBM3E stacks.  Classify 8-GiB allocations into the three classes, then
time the real fused step (N = 100M, r = 20, both branches, per-kernel) with U, V and the thin streams in chosen classes."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgd_tf_amd import _lib, placement  # noqa: E402

GiB = 1 << 30


def main():
    dev = torch.device("cuda:0")
    cls = placement.UVdArena
    lib = _lib.load()
    N, r = 100_000_000, 20
    chunk = 8 * GiB
    n = (chunk // 16) // (4 * r) // 64 * 64
    fo, fend = cls.sequential(n, r, ("U", "V"))
    ro, rend = cls.sequential(n, r, ("g", "v", "h"), fend)
    wo, wend = cls.sequential(n, r, ("d", "out", "ws"), rend)
    wo0, _ = cls.sequential(n, r, ("d", "out", "ws"), 0)

    def small_ms(fac, thin):
        where = {k: (fac, o) for k, o in list(fo.items()) + list(ro.items())}
        where.update({k: ((fac, o) if thin is None else (thin, wo0[k])) for k, o in wo.items()})
        return min(cls(n, r, dev, where).time_step(iters=4, final_only=True)[0] for _ in range(2))
    bufs = [torch.empty(chunk, dtype=torch.uint8, device=dev) for _ in range(int(os.environ.get("CHUNKS", "20")))]
    t_same = small_ms(bufs[0], None)
    klass = [0] * len(bufs)
    ref2 = None
    for i in range(1, len(bufs)):
        if small_ms(bufs[0], bufs[i]) < 0.95 * t_same:            # not class 0
            if ref2 is None:
                ref2 = i
            klass[i] = 1 if (i == ref2 or small_ms(bufs[ref2], bufs[i]) > 0.95 * t_same) else 2
    print("classes of the 8-GiB allocations in allocation order:", "".join(str(k) for k in klass), flush=True)
    pick = {c: [i for i, k in enumerate(klass) if k == c] for c in (0, 1, 2)}
    print({c: len(v) for c, v in pick.items()})
    st = torch.cuda.current_stream().cuda_stream

    def run(label, bu, bv, bthin, bread=None, bspare=None):
        """U in buffer bu, V in bv, written thin streams (d, out, ws) in bthin, read-only ones (g, v, h) in bread (default bthin);
        bspare: the updated factor is written OUT OF PLACE into that buffer (psgd_uvd_update_apply_oop_f32)"""
        bread = bthin if bread is None else bread
        sz = cls.region_bytes(N, r)
        where = {"U": (bufs[bu], 0), "V": (bufs[bv], 0)}
        used = {bu: sz["U"], bv: sz["V"]} if bu != bv else None
        assert bu != bv
        cur = {}
        for k, b in (("d", bthin), ("out", bthin), ("ws", bthin), ("g", bread), ("v", bread), ("h", bread)):
            assert b not in (bu, bv) and b != bspare
            o = cur.get(b, 0)
            where[k] = (bufs[b], o)
            cur[b] = (o + sz[k] + 255) // 256 * 256
        a = cls(N, r, dev, where)
        a.fill_initial(1.0)
        a.g.normal_(); a.v.normal_(); a.h.copy_(a.v).mul_(1.5)
        P = lambda t: t.data_ptr()

        def call(bu_):
            if bspare is not None:
                assert bspare not in (bu, bv)
                rc = lib.psgd_uvd_update_apply_oop_f32(P(a.U), P(a.V), P(a.d), P(a.v), P(a.h), P(a.g), P(a.out), bufs[bspare].data_ptr(), N, r,
                                                       0.0, 1.1754943508222875e-38, 0, bu_, P(a.ws), a.ws.numel(), st)
            else:
                rc = lib.psgd_uvd_update_apply_f32(P(a.U), P(a.V), P(a.d), P(a.v), P(a.h), P(a.g), P(a.out), N, r, 0.0, 1.1754943508222875e-38,
                                                   0, bu_, P(a.ws), a.ws.numel(), st)
            assert rc == 0
        call(1); call(0)
        res = []
        for br in (1, 0):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                call(br)
            e1.record()
            torch.cuda.synchronize()
            wall = e0.elapsed_time(e1) / 4
            lib.psgd_prof_enable(1)
            for _ in range(4):
                call(br)
            torch.cuda.synchronize()
            ks = []
            for slot in (3, 4, 2):
                tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
                lib.psgd_prof_collect(slot, ctypes.byref(tot), ctypes.byref(cnt))
                ks.append(tot.value / max(cnt.value, 1))
            lib.psgd_prof_enable(0)
            res.append((wall, ks))
        print("%-52s step %.3f / %.3f (%.3f) | s1 %.3f %.3f | s2 %.3f %.3f | fin %.3f %.3f" % (
            label, res[0][0], res[1][0], 0.5 * (res[0][0] + res[1][0]), res[0][1][0], res[1][1][0], res[0][1][1], res[1][1][1],
            res[0][1][2], res[1][1][2]), flush=True)
    c0, c1, c2 = pick[0], pick[1], pick[2]
    # (the out-of-place variants need a build with psgd_uvd_update_apply_oop_f32 -- an experiment of round 6 that was measured and not
    #  kept: profiles/r06_placement.txt section 7)
    if os.environ.get("OOP", "0") == "1" and hasattr(lib, "psgd_uvd_update_apply_oop_f32") and len(c0) >= 3 and len(c1) >= 2 and len(c2) >= 2:
        run("in place:  U c0, V c0, thin c1", c0[0], c0[1], c1[0])
        run("OOP -> c0: U c0, V c0, thin c1, spare c0", c0[0], c0[1], c1[0], bspare=c0[2])
        run("OOP -> c1: U c0, V c0, thin c1, spare c1", c0[0], c0[1], c1[0], bspare=c1[1])
        run("OOP -> c2: U c0, V c0, thin c1, spare c2", c0[0], c0[1], c1[0], bspare=c2[0])
        run("OOP -> c1: U c0, V c0, thin c2, spare c1", c0[0], c0[1], c2[0], bspare=c1[0])
        run("OOP -> c1: U c0, V c1, thin c2, spare c1 (V-branch: same region as V)", c0[0], c1[0], c2[0], bspare=c1[1])
        run("OOP -> c0: U c0, V c1, thin c2, spare c0", c0[0], c1[0], c2[0], bspare=c0[1])
        run("in place:  U c0, V c0, thin c1 (again)", c0[0], c0[1], c1[0])
        return
    if len(c0) >= 4:
        run("U c0, V c0, thin c0 (one class)", c0[0], c0[1], c0[2])
    if len(c0) >= 2 and c1:
        run("U c0, V c0, thin c1", c0[0], c0[1], c1[0])
    if len(c0) >= 2 and c2:
        run("U c0, V c0, thin c2", c0[0], c0[1], c2[0])
    if c1 and len(c0) >= 2:
        run("U c0, V c1, thin c0", c0[0], c1[0], c0[1])
    if len(c1) >= 2:
        run("U c0, V c1, thin c1", c0[0], c1[0], c1[1])
    if c1 and c2:
        run("U c0, V c1, thin c2", c0[0], c1[0], c2[0])
        run("U c1, V c2, thin c0", c1[0], c2[0], c0[0])
        run("U c2, V c0, thin c1", c2[0], c0[0], c1[0])
        if len(c0) >= 2:
            run("U c0, V c1, written thin c2, read thin c0", c0[0], c1[0], c2[0], c0[1])
        if len(c2) >= 2:
            run("U c0, V c1, written thin c2, read thin c2'", c0[0], c1[0], c2[0], c2[1])
        if len(c0) >= 2:
            run("U c0, V c0, written thin c1, read thin c2", c0[0], c0[1], c1[0], c2[0])


if __name__ == "__main__":
    main()
