"""Round 6: the three sweeps of precond_grad_UVd_math against their grid (psgd_set_tuning 10 + kind: 0 = sweep 1, 1 = sweep 2, 2 = sweep 3)
on a placed state, and the sparse-LU update / apply against the global blocks-per-CU cap of its sweeps."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import preconditioned_stochastic_gradient_descent as psgd  # noqa: E402
from psgd_tf_amd import _lib, placement  # noqa: E402


def main():
    N, r = int(os.environ.get("GS_N", 100_000_000)), int(os.environ.get("GS_R", 20))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    arena = placement.UVdArena.probe(N, r, dev)
    print("layout:", arena.info.get("layout"), flush=True)
    arena.fill_initial(1.0)
    arena.g.normal_()
    arena.install_workspace()

    def measure(iters=6):
        for _ in range(2):
            psgd.precond_grad_UVd_math(arena.U, arena.V, arena.d, arena.g, out=arena.out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            psgd.precond_grad_UVd_math(arena.U, arena.V, arena.d, arena.g, out=arena.out)
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) / iters
        lib.psgd_prof_enable(1)
        for _ in range(iters):
            psgd.precond_grad_UVd_math(arena.U, arena.V, arena.d, arena.g, out=arena.out)
        torch.cuda.synchronize()
        ks = []
        for slot in (0, 1, 2):
            tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
            lib.psgd_prof_collect(slot, ctypes.byref(tot), ctypes.byref(cnt))
            ks.append(tot.value / max(cnt.value, 1))
        lib.psgd_prof_enable(0)
        return "apply %.3f ms | s1 %.3f  s2 %.3f  s3 %.3f" % (wall, ks[0], ks[1], ks[2])
    print("%-30s %s" % ("default grids", measure()), flush=True)
    for kind, name in ((1, "sweep 2"), (2, "sweep 3"), (0, "sweep 1")):
        for grid in (256, 512, 768, 1024, 2048):
            lib.psgd_set_tuning(10 + kind, grid)
            print("%-30s %s" % ("%s on %d workgroups" % (name, grid), measure()), flush=True)
        lib.psgd_set_tuning(10 + kind, 0)
    print("%-30s %s" % ("default grids (again)", measure()), flush=True)


if __name__ == "__main__":
    main()
