"""Round 6, item 1, second experiment: the placement effect follows the POWER-OF-TWO PIECES a large hipMalloc is made of.

tools/r06_placement_scan.py --mode scan1d showed one sharp boundary at offset 32 GiB of a 48-GiB allocation (and 16 GiB of an 18.6-GiB
one): the VRAM manager backs a request with buddy blocks in descending powers of two, and a stream that is WRITTEN runs faster when it
lives in another block than the big read streams.  Here: allocations of chosen sizes (so that the block boundaries are known), a coarse
slide of the thin streams to see the boundaries, and candidate layouts that put U / V / the thin streams into different blocks or across
a boundary.  One JSON line per layout (same fields as the scan tool).

  python tools/r06_placement_blocks.py --out gpurun_out/r06_placement_blocks.jsonl
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from psgd_tf_amd import _lib  # noqa: E402

NAMES = ("U", "V", "d", "g", "v", "h", "out", "ws")
THIN = ("d", "g", "v", "h", "out", "ws")
SLOTS = (("update_s1", 3), ("update_s2", 4), ("final", 2))
G = 2**30


class Rig:
    def __init__(self, N, r, iters, fout):
        self.N, self.r, self.iters, self.fout = N, r, iters, fout
        self.dev = torch.device("cuda:0")
        self.lib = _lib.load()
        self.ws_bytes = int(self.lib.psgd_uvd_workspace_bytes(N, r))
        self.sizes = {"U": 4 * N * r, "V": 4 * N * r, "d": 4 * N, "g": 4 * N, "v": 4 * N, "h": 4 * N, "out": 4 * N, "ws": self.ws_bytes}
        self.scale = (1.0 / (N * r)) ** 0.5

    def thin_at(self, start, order=THIN):
        off, cur = {}, int(start) // 256 * 256
        for k in order:
            off[k] = cur
            cur = (cur + self.sizes[k] + 255) // 256 * 256
        return off

    def run(self, label, where):
        """where: name -> (buffer tensor (uint8), byte offset)"""
        t = {}
        for k in NAMES:
            buf, off = where[k]
            off = int(off) // 256 * 256
            assert 0 <= off and off + self.sizes[k] <= buf.numel(), (label, k, off / G)
            raw = buf[off:off + self.sizes[k]]
            t[k] = raw if k == "ws" else raw.view(torch.float32)
        spans = sorted((t[k].data_ptr(), t[k].data_ptr() + self.sizes[k]) for k in NAMES)
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), (label, "overlap")
        g = torch.Generator(device=self.dev).manual_seed(0)
        t["U"].normal_(generator=g).mul_(self.scale)
        t["V"].normal_(generator=g).mul_(self.scale)
        t["d"].fill_(1.0)
        t["g"].normal_(generator=g)
        t["v"].normal_(generator=g)
        t["h"].uniform_(-4.6, 4.6, generator=g).exp_().mul_(t["v"])
        ptr = {k: t[k].data_ptr() for k in NAMES}
        st = torch.cuda.current_stream().cuda_stream
        lib, N, r = self.lib, self.N, self.r

        def call(update_u):
            rc = lib.psgd_uvd_update_apply_f32(ptr["U"], ptr["V"], ptr["d"], ptr["v"], ptr["h"], ptr["g"], ptr["out"], N, r,
                                               0.01, 1.1754943508222875e-38, 0, update_u, ptr["ws"], self.ws_bytes, st)
            assert rc == 0, rc
        rec = {"label": label, "off": {k: int(where[k][1]) for k in NAMES}, "buf": {k: int(where[k][0].data_ptr()) for k in NAMES}}
        lib.psgd_prof_enable(0)
        call(1)
        call(0)
        for br, name in ((1, "U"), (0, "V")):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(self.iters):
                call(br)
            e1.record()
            torch.cuda.synchronize()
            rec["step_%s_ms" % name] = e0.elapsed_time(e1) / self.iters
            lib.psgd_prof_enable(1)
            for _ in range(self.iters):
                call(br)
            torch.cuda.synchronize()
            for kname, slot in SLOTS:
                tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
                lib.psgd_prof_collect(slot, ctypes.byref(tot), ctypes.byref(cnt))
                rec["%s_%s_ms" % (kname, name)] = tot.value / max(cnt.value, 1)
            lib.psgd_prof_enable(0)
        rec["step_alt_ms"] = 0.5 * (rec["step_U_ms"] + rec["step_V_ms"])
        self.fout.write(json.dumps(rec) + "\n")
        self.fout.flush()
        print("%-44s step %.3f %.3f (%.3f) | s1 %.3f %.3f | s2 %.3f %.3f | fin %.3f %.3f" % (
            label[:44], rec["step_U_ms"], rec["step_V_ms"], rec["step_alt_ms"], rec["update_s1_U_ms"], rec["update_s1_V_ms"],
            rec["update_s2_U_ms"], rec["update_s2_V_ms"], rec["final_U_ms"], rec["final_V_ms"]), flush=True)
        return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=100_000_000)
    ap.add_argument("--r", type=int, default=20)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--out", default="gpurun_out/r06_placement_blocks.jsonl")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    rig = Rig(args.N, args.r, args.iters, open(args.out, "a"))
    dev = rig.dev
    F = rig.sizes["U"]
    T = sum(rig.sizes[k] for k in THIN) + 8 * 256
    f, t = F / G, T / G

    def slab(gib):
        return torch.empty(int(gib * G), dtype=torch.uint8, device=dev)

    def lay(buf, U, V, thin, order=THIN):
        w = {"U": (buf, U * G), "V": (buf, V * G)}
        w.update({k: (buf, o) for k, o in rig.thin_at(thin * G, order).items()})
        return w

    # ---- 28 GiB = 16 + 8 + 4
    s = slab(28)
    x = 15.0
    while x + t <= 28:
        rig.run("S28 slide thin@%.1f (U 0, V 7.5)" % x, lay(s, 0, 7.5, x))
        x += 0.5
    rig.run("S28 U A | V A | thin C(24.5)", lay(s, 0, 7.5, 24.5))
    rig.run("S28 U A | V B(16) | thin C(24.5)", lay(s, 0, 16, 24.5))
    rig.run("S28 U A | V B(16) | thin A(8)", lay(s, 0, 16, 8))
    rig.run("S28 U A | thin B(16) | V B(19)", lay(s, 0, 19, 16))
    rig.run("S28 U A|B (10.5) V B|C (17.95) thin C", lay(s, 10.5, 10.5 + f + 0.001, 10.5 + 2 * f + 0.002))
    rig.run("S28 U A|B (12) V B|C (19.5) thin A(0)", lay(s, 12, 19.5, 0))
    rig.run("S28 thin A(0) U A (3) V B|C (19.5)", lay(s, 3, 19.5, 0))
    rig.run("S28 U A (8.5) V B|C(19.5) thin B(16)", lay(s, 8.5, 19.5, 16))
    del s
    torch.cuda.empty_cache()
    # ---- 20 GiB = 16 + 4: the smallest slab that holds everything with one boundary
    s = slab(20)
    rig.run("S20 U 0 V 7.5 thin B(16)", lay(s, 0, 7.5, 16))
    rig.run("S20 U 1 V 8.5 (ends 15.95) thin B(16)", lay(s, 1, 8.5, 16))
    for vstart in (9.5, 10.0, 10.5, 11.0, 11.5):
        # V across the boundary, U in front of it, the thin streams behind (D block@19 of the first scan)
        if vstart + f + t <= 20:
            rig.run("S20 U %.1f V %.1f A|B thin behind" % (vstart - f - 0.001, vstart), lay(s, vstart - f - 0.001, vstart, vstart + f + 0.001))
    del s
    torch.cuda.empty_cache()
    # ---- 24 GiB = 16 + 8
    s = slab(24)
    rig.run("S24 U A(0) V B(16) thin A(8)", lay(s, 0, 16, 8))
    rig.run("S24 U A(8) V B(16) thin A(0)", lay(s, 8, 16, 0))
    rig.run("S24 U A(0) V A(7.5) thin B(16)", lay(s, 0, 7.5, 16))
    rig.run("S24 U 3.05 V 10.5 A|B thin B(18)", lay(s, 3.05, 10.5, 18))
    rig.run("S24 U 5.05 V 12.5 A|B thin B(20)", lay(s, 5.05, 12.5, 20))
    del s
    torch.cuda.empty_cache()
    # ---- three separate 16-GiB allocations (each one block if the hypothesis holds)
    b1, b2, b3 = slab(16), slab(16), slab(16)
    w = {"U": (b1, 0), "V": (b2, 0)}
    w.update({k: (b3, o) for k, o in rig.thin_at(0).items()})
    rig.run("3x16: U | V | thin", w)
    w = {"U": (b1, 0), "V": (b1, 7.5 * G)}
    w.update({k: (b2, o) for k, o in rig.thin_at(0).items()})
    rig.run("3x16: U V | thin", w)
    w = {"U": (b1, 0), "V": (b2, 0)}
    w.update({k: (b1, o) for k, o in rig.thin_at(8 * G).items()})
    rig.run("3x16: U thin | V", w)
    # written thin streams (d, out, ws) apart from the read-only ones (g, v, h)
    w = {"U": (b1, 0), "V": (b1, 7.5 * G)}
    w.update({k: (b2, o) for k, o in rig.thin_at(0, ("d", "out", "ws")).items()})
    w.update({k: (b3, o) for k, o in rig.thin_at(0, ("g", "v", "h")).items()})
    rig.run("3x16: U V | d out ws | g v h", w)
    w = {"U": (b1, 0), "V": (b2, 0)}
    w.update({k: (b3, o) for k, o in rig.thin_at(0, ("d", "out", "ws")).items()})
    w.update({k: (b1, o) for k, o in rig.thin_at(8 * G, ("g", "v", "h")).items()})
    rig.run("3x16: U gvh | V | d out ws", w)
    del b1, b2, b3, w
    torch.cuda.empty_cache()
    # ---- exact-size separate allocations, thin first / last (what torch would hand a caller)
    for order in (("U", "V") + THIN, THIN + ("U", "V"), ("U",) + THIN + ("V",)):
        bufs = {k: torch.empty(rig.sizes[k], dtype=torch.uint8, device=dev) for k in order}
        rig.run("separate, order " + " ".join(order), {k: (bufs[k], 0) for k in NAMES})
        del bufs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
