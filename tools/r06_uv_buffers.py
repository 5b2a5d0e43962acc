"""Round 6: does it matter whether U and V share ONE 16-GiB allocation or sit in two 8-GiB ones of the same region?  (bench's arena:
Gram sweep 2.81 ms; tools/r06_three_ranks.py with separate 8-GiB buffers: 2.72.)  Thin streams in another region in every case."""
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
    sz = cls.region_bytes(N, r)
    n = (8 * GiB // 16) // (4 * r) // 64 * 64
    fo, fend = cls.sequential(n, r, ("U", "V"))
    ro, rend = cls.sequential(n, r, ("g", "v", "h"), fend)
    wo, wend = cls.sequential(n, r, ("d", "out", "ws"), rend)
    wo0, _ = cls.sequential(n, r, ("d", "out", "ws"), 0)

    def small_ms(fac, thin):
        where = {k: (fac, o) for k, o in list(fo.items()) + list(ro.items())}
        where.update({k: ((fac, o) if thin is None else (thin, wo0[k])) for k, o in wo.items()})
        return min(cls(n, r, dev, where).time_step(iters=4, final_only=True)[0] for _ in range(2))
    big = torch.empty(16 * GiB, dtype=torch.uint8, device=dev)
    bufs = [torch.empty(8 * GiB, dtype=torch.uint8, device=dev) for _ in range(14)]
    t_same = small_ms(big, None)
    same = [i for i, b in enumerate(bufs) if small_ms(big, b) > 0.95 * t_same]
    other = [i for i in range(len(bufs)) if i not in same]
    print("8-GiB buffers in the 16-GiB buffer's region:", same, " elsewhere:", other, flush=True)
    st = torch.cuda.current_stream().cuda_stream

    def run(label, where):
        a = cls(N, r, dev, where)
        a.fill_initial(1.0)
        a.g.normal_(); a.v.normal_(); a.h.copy_(a.v).mul_(1.5)
        P = lambda t: t.data_ptr()

        def call(bu_):
            rc = lib.psgd_uvd_update_apply_f32(P(a.U), P(a.V), P(a.d), P(a.v), P(a.h), P(a.g), P(a.out), N, r, 0.0, 1.1754943508222875e-38,
                                               0, bu_, P(a.ws), a.ws.numel(), st)
            assert rc == 0
        call(1); call(0)
        res = []
        for br in (1, 0):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                call(br)
            e1.record()
            torch.cuda.synchronize()
            wall = e0.elapsed_time(e1) / 6
            lib.psgd_prof_enable(1)
            for _ in range(6):
                call(br)
            torch.cuda.synchronize()
            ks = []
            for slot in (3, 4, 2):
                tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
                lib.psgd_prof_collect(slot, ctypes.byref(tot), ctypes.byref(cnt))
                ks.append(tot.value / max(cnt.value, 1))
            lib.psgd_prof_enable(0)
            res.append((wall, ks))
        print("%-64s step %.3f / %.3f (%.3f) | s1 %.3f %.3f | s2 %.3f %.3f | fin %.3f %.3f" % (
            label, res[0][0], res[1][0], 0.5 * (res[0][0] + res[1][0]), res[0][1][0], res[1][1][0], res[0][1][1], res[1][1][1],
            res[0][1][2], res[1][1][2]), flush=True)
    if not other or len(same) < 2:
        print("not enough buffers of both kinds")
        return
    thin = {k: (bufs[other[0]], o) for k, o in cls.sequential(N, r, cls.THIN)[0].items()}
    F = (sz["U"] + 255) // 256 * 256
    for rep in range(2):
        run("U, V back to back in the 16-GiB buffer", dict(thin, U=(big, 0), V=(big, F)))
        run("U at 0, V at 8 GiB of the 16-GiB buffer", dict(thin, U=(big, 0), V=(big, 8 * GiB)))
        run("U, V in two 8-GiB buffers of the same region", dict(thin, U=(bufs[same[0]], 0), V=(bufs[same[1]], 0)))
        run("U in the 16-GiB buffer, V in an 8-GiB buffer of its region", dict(thin, U=(big, 0), V=(bufs[same[0]], 0)))
        run("U at 256 MiB, V at 8 GiB + 64 MiB of the 16-GiB buffer", dict(thin, U=(big, 256 << 20), V=(big, 8 * GiB + (64 << 20))))


if __name__ == "__main__":
    main()
