"""Round 6, placement: walk through ALL of the device's free memory in 4-GiB allocations (each kept) and time the small problem's last
sweep with its factors in the first 16-GiB buffer and its written thin streams in allocation i -- the region map of the whole VRAM in
the order the allocator hands it out.  python tools/r06_region_walk.py [chunk_gib]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgd_tf_amd import placement  # noqa: E402

GiB = 1 << 30


def main():
    chunk = int(float(sys.argv[1]) * GiB) if len(sys.argv) > 1 else 4 * GiB
    dev = torch.device("cuda:0")
    cls = placement.UVdArena
    r = 20
    free, total = torch.cuda.mem_get_info(dev)
    print("free %.1f GiB of %.1f" % (free / GiB, total / GiB), flush=True)
    A = torch.empty(16 * GiB, dtype=torch.uint8, device=dev)
    n = (16 * GiB // 16) // (4 * r) // 64 * 64
    fo, fend = cls.sequential(n, r, ("U", "V"))
    ro, rend = cls.sequential(n, r, ("g", "v", "h"), fend)
    wo, wend = cls.sequential(n, r, ("d", "out", "ws"), rend)
    wo0, _ = cls.sequential(n, r, ("d", "out", "ws"), 0)

    def small_ms(fac, thin, thin_off=0):
        where = {k: (fac, o) for k, o in list(fo.items()) + list(ro.items())}
        where.update({k: ((fac, o) if thin is None else (thin, thin_off + wo0[k])) for k, o in wo.items()})
        return min(cls(n, r, dev, where).time_step(iters=4, final_only=True)[0] for _ in range(2))
    t_same = small_ms(A, None)
    print("factors in A (16 GiB at %#x), written streams inside A: %.4f ms" % (A.data_ptr(), t_same), flush=True)
    # inside A: its second half
    print("   ... at A + 8 GiB: %.4f   at A + 15 GiB: %.4f" % (small_ms(A, A, 8 * GiB), small_ms(A, A, 15 * GiB)), flush=True)
    bufs, line = [], []
    while True:
        free, _ = torch.cuda.mem_get_info(dev)
        if free < chunk + 3 * GiB:
            break
        try:
            bufs.append(torch.empty(chunk, dtype=torch.uint8, device=dev))
        except RuntimeError:
            break
        t = small_ms(A, bufs[-1])
        line.append(t)
        print("alloc %3d  (+%5.1f GiB after A, va %#x): %.4f ms  %s" % (len(bufs), len(bufs) * chunk / GiB, bufs[-1].data_ptr(), t,
                                                                   "OTHER REGION" if t < 0.95 * t_same else ""), flush=True)
    other = [i for i, t in enumerate(line) if t < 0.95 * t_same]
    print("allocations in another region than A: %s of %d" % ([i + 1 for i in other], len(line)))
    # factors moved into the other region (if any): is A's region then the fast place for the written streams?
    if other and chunk >= 4 * GiB:
        B = bufs[other[0]]
        print("factors in allocation %d, written streams: same buffer %.4f, in A %.4f, in allocation 1 %.4f" % (
            other[0] + 1, small_ms(B, None), small_ms(B, A), small_ms(B, bufs[0])))


if __name__ == "__main__":
    main()
