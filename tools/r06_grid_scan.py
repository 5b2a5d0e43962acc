"""Round 6: the sweeps of the fused step against their GRID (workgroups), per kernel, on the placed state -- update sweep 2 ran faster
with ONE workgroup per CU than with three (tools/uvd_timing.py --bpc 1).  psgd_set_tuning(10 + kind, workgroups): kind 4 = the Gram
sweep, 7 = fused sweep 2, 8 = the last sweep."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgd_tf_amd import _lib, placement  # noqa: E402

TINY = 1.1754943508222875e-38


def main():
    N, r = int(os.environ.get("GS_N", 100_000_000)), int(os.environ.get("GS_R", 20))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    log = []
    arena = placement.UVdArena.probe(N, r, dev, log=log) if os.environ.get("GS_PLACE", "1") == "1" else placement.UVdArena.packed(N, r, dev)
    print("layout:", arena.info.get("layout"), flush=True)
    arena.fill_initial(1.0)
    arena.g.normal_(); arena.v.normal_(); arena.h.copy_(arena.v).mul_(1.5)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()

    def call(bu):
        rc = lib.psgd_uvd_update_apply_f32(P(arena.U), P(arena.V), P(arena.d), P(arena.v), P(arena.h), P(arena.g), P(arena.out), N, r,
                                           0.0, TINY, 0, bu, P(arena.ws), arena.ws.numel(), st)
        assert rc == 0

    def measure(iters=4):
        call(1); call(0)
        out = []
        for br in (1, 0):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                call(br)
            e1.record()
            torch.cuda.synchronize()
            wall = e0.elapsed_time(e1) / iters
            lib.psgd_prof_enable(1)
            for _ in range(iters):
                call(br)
            torch.cuda.synchronize()
            ks = []
            for slot in (3, 4, 2):
                tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
                lib.psgd_prof_collect(slot, ctypes.byref(tot), ctypes.byref(cnt))
                ks.append(tot.value / max(cnt.value, 1))
            lib.psgd_prof_enable(0)
            out.append((wall, ks))
        return out
    base = measure()
    fmt = lambda res: "step %.3f / %.3f | s1 %.3f %.3f | s2 %.3f %.3f | fin %.3f %.3f" % (
        res[0][0], res[1][0], res[0][1][0], res[1][1][0], res[0][1][1], res[1][1][1], res[0][1][2], res[1][1][2])
    print("%-28s %s" % ("default grids", fmt(base)), flush=True)
    for kind, name in ((7, "sweep 2"), (8, "last sweep"), (4, "Gram sweep")):
        for grid in [int(x) for x in os.environ.get("GS_GRIDS", "64,128,192,256,320,384,512,768,1024,2048").split(",")]:
            lib.psgd_set_tuning(10 + kind, grid)
            print("%-28s %s" % ("%s on %d workgroups" % (name, grid), fmt(measure())), flush=True)
        lib.psgd_set_tuning(10 + kind, 0)
    print("%-28s %s" % ("default grids (again)", fmt(measure())), flush=True)


if __name__ == "__main__":
    main()
