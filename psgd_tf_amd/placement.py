"""Where the UVd state lives in HBM: one owner for U, V, d, the workspace and the output (psgd.py:600, :614, :688-690).

Why.  The fused step's sweeps move exactly their algorithmic bytes (PMC traffic = 1.001 x), yet the same binary runs its dominant
kernel in 4.6 or 5.05 ms depending on where its streams sit in PHYSICAL memory (profiles/r06_placement.txt).  What the round-6
scans found on MI355X:
  * offsets inside one physically uniform allocation do not matter at all (gaps from 256 B to 256 MiB, any order of the eight
    regions: +-0.3 %);
  * a 48-GiB hipMalloc is backed by a 32-GiB and a 16-GiB block (the VRAM manager hands out powers of two, largest first) that come
    from two different regions of the address map, and a stream that is WRITTEN (d, out, nablaD, the rewritten factor) runs
    faster when it lives in the other region than the big read streams: final sweep 3.38 -> 2.91 ms with d / out / nablaD behind
    the boundary, update sweep 2 5.05 -> 4.88 ms; read-only streams (g, v, h) do not care;
  * a factor that lies ACROSS the boundary is rewritten fastest of all (4.62 ms), so the best layout found is
    U in front of the boundary, V across it (about a quarter behind), every thin stream behind: 11.2 -> 10.4-10.6 ms per step;
  * 16-, 20-, 24- and 28-GiB allocations and separate exact-size allocations all came out of ONE region (flat 11.2 ms).
Nothing of this can be asked of the driver, so it is MEASURED: `UVdArena.probe` allocates the two-block slab when the device has
room for it, finds the boundary by sliding the three written thin streams over the slab and timing the last sweep, times the real
fused step (both branches) on the candidate layouts and on the packed exact-size allocation, and keeps the fastest -- the packed
one if nothing is gained (the slab is freed again).  Results are bit-identical whatever is chosen: only addresses change.

Who uses it.  `UVd(..., placement="probe")` (the optimizer owns its state: psgd.py:688-690) and bench.py's headline workload.
Callers of the functional API get the same through `uvd_placed_state`: U, V, d to pass to update_precond_UVd_math_ /
precond_grad_UVd_math, with the workspace of that (N, r, stream) installed in the arena and `arena.out` for the `out=` argument of
update_precond_UVd_math_and_precond_grad.
"""
import math

import torch

from . import _lib

GiB = 1 << 30
_ALIGN = 256
_TINY = 1.1754943508222875e-38


def _up(x, a=_ALIGN):
    return -(-int(x) // a) * a


class UVdArena:
    """U, V, d, out, the workspace and (optionally used) input staging vectors g, v, h as views of ONE device allocation."""

    NAMES = ("U", "V", "d", "out", "ws", "g", "v", "h")

    def __init__(self, N, r, device, slab, offsets, info=None):
        self.N, self.r, self.device, self.slab, self.offsets = int(N), int(r), torch.device(device), slab, dict(offsets)
        self.info = info or {}
        sz = self.region_bytes(N, r)
        view = lambda k: slab[self.offsets[k]:self.offsets[k] + sz[k]]
        self.U = view("U").view(torch.float32).view(N, r)
        self.V = view("V").view(torch.float32).view(N, r)
        self.d, self.out, self.g, self.v, self.h = (view(k).view(torch.float32).view(N, 1) for k in ("d", "out", "g", "v", "h"))
        self.ws = view("ws")

    @staticmethod
    def region_bytes(N, r):
        ws = int(_lib.load().psgd_uvd_workspace_bytes(N, r))
        if ws < 0:
            _lib.check(ws, "psgd_uvd_workspace_bytes")
        return {"U": 4 * N * r, "V": 4 * N * r, "d": 4 * N, "out": 4 * N, "ws": ws, "g": 4 * N, "v": 4 * N, "h": 4 * N}

    # ---------------------------------------------------------------- layouts
    @classmethod
    def packed_offsets(cls, N, r):
        sz, off, cur = cls.region_bytes(N, r), {}, 0
        for k in cls.NAMES:
            off[k] = cur
            cur = _up(cur + sz[k])
        return off, cur

    @classmethod
    def boundary_offsets(cls, N, r, boundary, slab_bytes, straddle=0.25):
        """U in front of `boundary` (a byte offset of the slab), V across it with the fraction `straddle` behind (0: V ends at the
        boundary), every thin stream behind V.  None if it does not fit."""
        sz = cls.region_bytes(N, r)
        F = _up(sz["U"])
        v0 = _up(boundary - int((1.0 - straddle) * F)) if straddle > 0 else (boundary - F) // _ALIGN * _ALIGN
        off = {"U": v0 - F, "V": v0}
        cur = max(_up(v0 + F), _up(boundary))
        for k in ("d", "out", "ws", "g", "v", "h"):
            off[k] = cur
            cur = _up(cur + sz[k])
        if off["U"] < 0 or cur > slab_bytes:
            return None
        return off

    def install_workspace(self, stream=None):
        """make the sweeps of this (device, N, r, stream) use the arena's workspace region (the product module's cache)"""
        from . import preconditioned_stochastic_gradient_descent as _psgd
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _psgd._ws_cache.put((idx, self.N, self.r, st), self.ws)

    # ---------------------------------------------------------------- timing
    def time_step(self, iters=3, final_only=False):
        """ms per fused update -> apply call on this arena's regions, branches (update U, update V) separately: HIP events on the
        current stream.  The state is filled with the reference's initial values first (psgd.py:687-690) and v, h, g with
        noise; step = 0 keeps it there.  final_only: the last sweep alone (psgd_uvd_fused_final_f32)."""
        lib, N, r = _lib.load(), self.N, self.r
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.fill_initial(1.0)
        self.g.normal_()
        self.v.normal_()
        self.h.copy_(self.v).mul_(1.5)
        P = lambda t: t.data_ptr()

        def call(bu):
            rc = lib.psgd_uvd_update_apply_f32(P(self.U), P(self.V), P(self.d), P(self.v), P(self.h), P(self.g), P(self.out), N, r,
                                               0.0, _TINY, 0, bu, P(self.ws), self.ws.numel(), st)
            _lib.check(rc, "psgd_uvd_update_apply_f32")

        def final():
            rc = lib.psgd_uvd_fused_final_f32(P(self.U), P(self.V), P(self.d), P(self.g), P(self.out), N, r, 0.0, _TINY,
                                              P(self.ws), self.ws.numel(), st)
            _lib.check(rc, "psgd_uvd_fused_final_f32")
        call(1)
        call(0)
        out = []
        for fn in ((final,) if final_only else (lambda: call(1), lambda: call(0))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize(self.device)
            out.append(e0.elapsed_time(e1) / iters)
        return out

    def fill_initial(self, init_scale=1.0, generator=None):
        """psgd.py:687-690 on the arena's state: U, V ~ N(0, 1) / sqrt(N r), d = init_scale"""
        sc = (1.0 / (self.N * self.r)) ** 0.5
        self.U.normal_(generator=generator).mul_(sc)
        self.V.normal_(generator=generator).mul_(sc)
        self.d.fill_(float(init_scale))

    # ---------------------------------------------------------------- the probe
    @classmethod
    def packed(cls, N, r, device):
        off, total = cls.packed_offsets(N, r)
        slab = torch.empty(total, dtype=torch.uint8, device=device)
        return cls(N, r, device, slab, off, {"layout": "packed", "slab_bytes": total})

    @classmethod
    def probe(cls, N, r, device, two_block_gib=None, min_gain=0.01, log=None):
        """The fastest of: the packed exact-size slab, and layouts around the block boundary of a 2^k + 2^(k-1) slab (measured).
        Falls back to the packed slab when the device has no room for the larger one, when no boundary shows, or when nothing
        gains at least `min_gain`.  `log`: a list that receives one dict per timed candidate."""
        device = torch.device(device)
        say = (lambda rec: log.append(rec)) if log is not None else (lambda rec: None)
        sz = cls.region_bytes(N, r)
        need = sum(_up(v) for v in sz.values())
        base = cls.packed(N, r, device)
        tb = base.time_step()
        best, best_ms = base, 0.5 * (tb[0] + tb[1])
        say({"layout": "packed", "slab_gib": need / GiB, "step_U_ms": tb[0], "step_V_ms": tb[1]})
        base.info.update(step_U_ms=tb[0], step_V_ms=tb[1], candidates=1)
        # the two-block slab: 2^k + 2^(k-1) with 2^k >= what sits in front of the boundary (U and three quarters of V); the VRAM
        # manager only has to leave its uniform top region for blocks of 32 GiB and more, so k >= 35
        F = _up(sz["U"])
        k = max(35, math.ceil(math.log2(max(1.75 * F, 1))))
        slab_bytes = int(two_block_gib * GiB) if two_block_gib else (1 << k) + (1 << (k - 1))
        free, _total = torch.cuda.mem_get_info(device)
        if need + F > slab_bytes or free < slab_bytes + (2 << 30):
            base.info["note"] = "no room for the two-block slab (%.0f GiB): packed" % (slab_bytes / GiB)
            return base
        try:
            slab = torch.empty(slab_bytes, dtype=torch.uint8, device=device)
        except RuntimeError:                                   # (out of memory: another tenant holds the device)
            base.info["note"] = "two-block slab allocation failed: packed"
            return base
        # 1. where is the boundary?  The last sweep with d / out / nablaD slid over the slab behind U and V (coarse, then fine)
        thin3 = _up(sz["d"]) + _up(sz["out"]) + _up(sz["ws"])
        rest = _up(sz["g"]) + _up(sz["v"]) + _up(sz["h"])

        def final_ms(x):
            off = {"U": 0, "V": F, "g": 2 * F, "v": 2 * F + _up(sz["g"]), "h": 2 * F + _up(sz["g"]) + _up(sz["v"]),
                   "d": x, "out": x + _up(sz["d"]), "ws": x + _up(sz["d"]) + _up(sz["out"])}
            return cls(N, r, device, slab, off).time_step(iters=3, final_only=True)[0]
        lo, hi = 2 * F + rest, slab_bytes - thin3
        boundary, scan = None, []
        if hi > lo:
            n = 13
            xs = [_up(lo + (hi - lo) * i / (n - 1)) if i < n - 1 else hi // _ALIGN * _ALIGN for i in range(n)]
            ts = [final_ms(x) for x in xs]
            scan = [(x / GiB, t) for x, t in zip(xs, ts)]
            drop = max(range(1, n), key=lambda i: ts[i - 1] - ts[i])
            if ts[drop - 1] - ts[drop] > 0.03 * ts[drop - 1] and min(ts[drop:]) < 0.97 * max(ts[:drop]):
                a, b = xs[drop - 1], xs[drop]                          # the streams' START crossed the boundary in (a, b]
                for _ in range(5):                                     # bisect to ~1/32 of the coarse step
                    m = _up((a + b) // 2)
                    if final_ms(m) < 0.5 * (ts[drop - 1] + ts[drop]):
                        b = m
                    else:
                        a = m
                boundary = b
                # the streams' START was timed: the drop is half done when about half of the 1.2 GiB of written streams is behind the
                # boundary, so the boundary itself is a little above b.  Blocks are powers of two: take the most aligned multiple of
                # 64 MiB in [b - 0.25 GiB, b + 0.9 GiB]
                cands = [m for m in range(((b - GiB // 4) >> 26) << 26, b + (9 * GiB) // 10, 1 << 26) if m > 0]
                if cands:
                    boundary = max(cands, key=lambda m: (m & -m, -abs(m - b)))
        say({"layout": "boundary scan (final sweep ms vs offset of d/out/nablaD, GiB)", "scan": scan,
             "boundary_gib": None if boundary is None else boundary / GiB})
        if boundary is None:
            del slab
            torch.cuda.empty_cache()
            base.info.update(note="no block boundary with an effect in a %.0f-GiB slab: packed" % (slab_bytes / GiB), scan=scan)
            return base
        # 2. the candidate layouts around it, on the real step
        n_cand = 1
        for straddle in (0.25, 0.0, 0.4):
            off = cls.boundary_offsets(N, r, boundary, slab_bytes, straddle)
            if off is None:
                continue
            cand = cls(N, r, device, slab, off, {"layout": "two-block slab, V %d %% behind the boundary" % round(100 * straddle),
                                                 "slab_bytes": slab_bytes, "boundary_gib": boundary / GiB})
            t = cand.time_step()
            n_cand += 1
            say({"layout": cand.info["layout"], "slab_gib": slab_bytes / GiB, "boundary_gib": boundary / GiB,
                 "step_U_ms": t[0], "step_V_ms": t[1], "offsets_gib": {k_: v / GiB for k_, v in off.items()}})
            cand.info.update(step_U_ms=t[0], step_V_ms=t[1])
            if 0.5 * (t[0] + t[1]) < best_ms * (1.0 - (min_gain if best is base else 0.0)):
                best, best_ms = cand, 0.5 * (t[0] + t[1])
        best.info.update(candidates=n_cand, packed_step_ms=0.5 * (tb[0] + tb[1]), scan=scan)
        if best is base:
            del slab
            torch.cuda.empty_cache()
            base.info["note"] = "no layout of the two-block slab gained %.0f %%: packed" % (100 * min_gain)
        else:
            del base
            torch.cuda.empty_cache()
        return best


def uvd_placed_state(N, r, device, preconditioner_init_scale=1.0, placement="probe", generator=None, log=None):
    """(U, V, d, arena) for the functional API: the initial state of psgd.py:687-690 in a placed arena whose workspace region is
    installed for the current stream.  placement: "probe" (UVdArena.probe) or "packed"."""
    if placement not in ("probe", "packed"):
        raise ValueError("placement: 'probe' or 'packed', got %r" % (placement,))
    arena = UVdArena.probe(N, r, device, log=log) if placement == "probe" else UVdArena.packed(N, r, device)
    arena.fill_initial(preconditioner_init_scale, generator)
    arena.install_workspace()
    return arena.U, arena.V, arena.d, arena
