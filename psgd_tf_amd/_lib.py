"""ctypes binding of libpsgd_hip.so (the C ABI declared in include/psgd_hip.h).

The library is built in-tree by ``build_extension()`` (hipcc, --offload-arch=gfx950)
and loaded lazily.  There is no fallback: if the shared object is missing or a
call returns an error code, a PsgdHipError is raised.
"""
import ctypes
import os
import subprocess
import threading

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("PSGD_HIP_LIB", os.path.join(_CSRC, "libpsgd_hip.so"))   # override: build experiments only

PSGD_OK = 0
PSGD_ERR_BAD_ARG, PSGD_ERR_RANK, PSGD_ERR_WORKSPACE, PSGD_ERR_ALIGN, PSGD_ERR_LAUNCH, PSGD_ERR_SHAPE = -1, -2, -3, -4, -5, -6
PSGD_ABI_VERSION = 7       # must equal include/psgd_hip.h (bumped on every incompatible change of symbols or workspace layout)
PSGD_WS_SUMS_F64 = 0
PSGD_WS_MAX_F32 = 1
PSGD_WS_SEND_F64 = 2
UVD_MAX_RANK = 32
SPLU_MAX_RANK = 64       # PSGD_SPLU_MAX_RANK: the native sparse-LU entry points (round 5)


class PsgdHipError(RuntimeError):
    pass


_c_f32p = ctypes.c_void_p   # device pointers travel as integers
_c_ws = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_flt = ctypes.c_float
_strm = ctypes.c_void_p

# name -> (restype, argtypes); kept in one table so that tests can check that every
# symbol declared in include/psgd_hip.h is bound and exported.
SIGNATURES = {
    "psgd_abi_version": (_int, []),
    "psgd_error_string": (ctypes.c_char_p, [_int]),
    "psgd_set_tuning": (_int, [_int, _int]),
    "psgd_prof_enable": (_int, [_int]),
    "psgd_prof_collect": (_int, [_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_int)]),
    "psgd_uvd_workspace_bytes": (_i64, [_i64, _int]),
    "psgd_uvd_ws_region": (_int, [_int, _int, _i64, _int, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "psgd_uvd_fold_gathered_f64": (_int, [_int, _c_f32p, _int, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_splu_fold_gathered_f64": (_int, [_int, _c_f32p, _int, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_apply_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_apply_sweep1_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_apply_sweep2_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_apply_sweep3_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _int, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_update_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _flt, _flt, _int, _int,
                                   _c_ws, _i64, _strm]),
    "psgd_uvd_update_apply_f32": (_int, [_c_f32p] * 7 + [_i64, _int, _flt, _flt, _int, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_update_sweep2_fused_f32": (_int, [_c_f32p] * 6 + [_i64, _int, _flt, _flt, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_fused_post_f32": (_int, [_i64, _int, _flt, _flt, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_fused_final_f32": (_int, [_c_f32p] * 5 + [_i64, _int, _flt, _flt, _c_ws, _i64, _strm]),
    "psgd_uvd_balance_max_f32": (_int, [_c_f32p, _c_f32p, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_balance_scale_f32": (_int, [_c_f32p, _c_f32p, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_update_sweep1_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _c_ws, _i64,
                                          _strm]),
    "psgd_uvd_update_sweep2_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _flt, _flt, _int,
                                          _c_ws, _i64, _strm]),
    "psgd_uvd_update_sweep3_f32": (_int, [_c_f32p, _i64, _int, _flt, _flt, _c_ws, _i64, _strm]),
    "psgd_uvd_ipuvt_matvec_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_ipuvt_matvec_cols_f32": (_int, [_c_f32p, _c_f32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                              _int, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_apply_cols_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                       _int, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_gram_wide_scratch_bytes": (_i64, [_i64, _int]),
    "psgd_uvd_gram_wide_f32": (_int, [_c_f32p] * 5 + [_i64, _int, _c_f32p, _c_ws, _i64, _strm]),
    "psgd_uvd_wide_scratch_bytes": (_i64, [_i64, _int]),
    "psgd_uvd_wide_apply_cols_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                            _int, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_wide_colsums_f32": (_int, [_c_f32p, ctypes.POINTER(ctypes.c_void_p), _int, _c_f32p, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_wide_axpy_cols_f32": (_int, [_c_f32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _int, _c_f32p,
                                           _i64, _int, _strm]),
    "psgd_uvd_wide_rank2_update_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _strm]),
    "psgd_uvd_wide_update_scratch_bytes": (_i64, [_i64, _int]),
    "psgd_uvd_wide_update_apply_scratch_bytes": (_i64, [_i64, _int]),
    "psgd_uvd_wide_update_apply_f32": (_int, [_c_f32p] * 7 + [_i64, _int, ctypes.c_float, ctypes.c_float, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_wide_update_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, ctypes.c_float, ctypes.c_float, _int,
                                        _c_ws, _i64, _strm]),
    "psgd_uvd_colsums_f32": (_int, [_c_f32p, ctypes.POINTER(ctypes.c_void_p), _int, _c_f32p, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_axpy_cols_f32": (_int, [_c_f32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _int, _c_f32p,
                                      _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_rank2_update_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_colsums_ld_f32": (_int, [_c_f32p, _i64, ctypes.POINTER(ctypes.c_void_p), _int, _c_f32p, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_axpy_cols_ld_f32": (_int, [_c_f32p, _i64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _int, _c_f32p,
                                         _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_rank2_update_ld_f32": (_int, [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _c_ws, _i64, _strm]),
    "psgd_uvd_update_sweep1_ld_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _i64, _int, _c_ws, _i64,
                                             _strm]),
    "psgd_splu_workspace_bytes": (_i64, [_i64, _int]),
    "psgd_splu_apply_f32": (_int, [_c_f32p] * 6 + [_i64, _int, _c_ws, _i64, _strm]),
    "psgd_splu_update_f32": (_int, [_c_f32p] * 10 + [_i64, _int, ctypes.c_float, ctypes.c_float, _c_ws, _i64, _strm]),
    "psgd_splu_ws_region": (_int, [_int, _int, _i64, _int, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "psgd_splu_stage1_f32": (_int, [_c_f32p] * 2 + [_i64, _int, _c_ws, _i64, _strm]),
    "psgd_splu_apply_stage2_f32": (_int, [_c_f32p] * 6 + [_i64, _int, _c_ws, _i64, _strm]),
    "psgd_splu_apply_stage3_f32": (_int, [_c_f32p] * 5 + [_i64, _int, _c_ws, _i64, _strm]),
    "psgd_splu_update_stage2_f32": (_int, [_c_f32p] * 6 + [_i64, _int, _c_ws, _i64, _strm]),
    "psgd_splu_update_stage3_f32": (_int, [_c_f32p] * 6 + [_i64, _int, _c_ws, _i64, _strm]),
    "psgd_splu_update_stage4_f32": (_int, [_c_f32p] * 10 + [_i64, _int, ctypes.c_float, ctypes.c_float, _int, _c_ws, _i64, _strm]),
    "psgd_kron_dd_workspace_bytes": (_i64, [_int, _int]),
    "psgd_kron_set_tuning": (_int, [_int, _int]),
    "psgd_kron_dd_workspace_bytes_batched": (_i64, [ctypes.POINTER(_int), ctypes.POINTER(_int), _int]),
    "psgd_kron_dd_apply_batched_f32": (_int, [ctypes.POINTER(ctypes.c_void_p)] * 4 + [ctypes.POINTER(_int)] * 2 +
                                       [_int, _c_ws, _i64, _strm]),
    "psgd_kron_dd_update_batched_f32": (_int, [ctypes.POINTER(ctypes.c_void_p)] * 6 + [ctypes.POINTER(_int)] * 2 +
                                        [_int, _flt, _flt, _c_ws, _i64, _strm]),
    "psgd_kron_dd_apply_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _int, _int, _c_ws, _i64, _strm]),
    "psgd_kron_dd_prepare_f32": (_int, [_c_f32p, _c_f32p, _int, _int, _c_ws, _i64, _strm]),
    "psgd_kron_dd_apply_prepared_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _int, _int, _c_ws, _i64, _strm]),
    "psgd_kron_dd_prepare_batched_f32": (_int, [ctypes.POINTER(ctypes.c_void_p)] * 2 + [ctypes.POINTER(_int)] * 2 +
                                         [_int, _c_ws, _i64, _strm]),
    "psgd_kron_dd_apply_prepared_batched_f32": (_int, [ctypes.POINTER(ctypes.c_void_p)] * 4 + [ctypes.POINTER(_int)] * 2 +
                                                [_int, _c_ws, _i64, _strm]),
    "psgd_kron_sparse_workspace_bytes": (_i64, [_int, _int, _int]),
    "psgd_kron_ds_update_f32": (_int, [_c_f32p] * 4 + [_i64, _i64, _c_f32p, _c_f32p, _int, _int, _flt, _flt, _c_ws, _i64, _strm]),
    "psgd_kron_ds_apply_f32": (_int, [_c_f32p] * 3 + [_i64, _i64, _c_f32p, _int, _int, _c_ws, _i64, _strm]),
    "psgd_kron_nd_update_f32": (_int, [_c_f32p] * 4 + [_i64, _i64, _c_f32p, _c_f32p, _int, _int, _flt, _flt, _c_ws, _i64, _strm]),
    "psgd_kron_nd_apply_f32": (_int, [_c_f32p] * 3 + [_i64, _i64, _c_f32p, _int, _int, _c_ws, _i64, _strm]),
    "psgd_kron_ns_update_f32": (_int, [_c_f32p] * 4 + [_i64, _i64, _c_f32p, _c_f32p, _int, _int, _flt, _flt, _c_ws, _i64, _strm]),
    "psgd_kron_ns_apply_f32": (_int, [_c_f32p] * 3 + [_i64, _i64, _c_f32p, _int, _int, _c_ws, _i64, _strm]),
    "psgd_kron_bf16_set_tuning": (_int, [_int, _int]),
    "psgd_kron_dd_workspace_bytes_bf16": (_i64, [_int, _int]),
    "psgd_kron_bf16_handoff_timeouts": (_int, [_c_ws, _int, _int]),
    "psgd_kron_bf16_handoff_reset": (_int, [_c_ws, _int, _int, _strm]),
    "psgd_kron_bf16_handoff_counter_offset": (_i64, [_int, _int]),
    "psgd_kron_dd_update_workspace_bytes_bf16": (_i64, [_int, _int]),
    "psgd_kron_dd_update_bf16": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _int, _int, _flt, _flt,
                                        _c_ws, _i64, _strm]),
    "psgd_kron_dd_apply_bf16": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _int, _int, _c_ws, _i64, _strm]),
    "psgd_kron_bf16_prepare_factors": (_int, [_c_f32p, _c_f32p, _int, _int, _c_ws, _i64, _strm]),
    "psgd_kron_dd_apply_bf16_prepared": (_int, [_c_f32p, _c_f32p, _int, _int, _c_ws, _i64, _strm]),
    "psgd_kron_dd_apply_direct_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _int, _int, _c_ws, _i64, _strm]),
    "psgd_kron_dd_apply_direct_distinct": (_int, [_int, _int]),
    "psgd_kron_dd_update_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _int, _int, _flt, _flt,
                                       _c_ws, _i64, _strm]),
}

_lib = None
_lock = threading.Lock()


def build_extension(force=False, jobs=None, verbose=False):
    """Compile libpsgd_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    jobs = jobs or min(8, os.cpu_count() or 1)
    cmd = ["make", "-C", _CSRC, "-j%d" % jobs]
    if force:
        subprocess.run(["make", "-C", _CSRC, "clean"], check=True, capture_output=not verbose)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise PsgdHipError("building libpsgd_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


def load():
    """Return the loaded ctypes library, loading (never building) it on first use."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PsgdHipError(
                "HIP extension %s is missing; run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C psgd_tf_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
        # torch ships its own libamdhip64.so (same SONAME); import it first so that the
        # extension binds to the HIP runtime that owns torch's device memory and streams.
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.psgd_abi_version() != PSGD_ABI_VERSION:
            raise PsgdHipError("libpsgd_hip.so ABI version mismatch: %s reports %d, this binding needs %d (rebuild with "
                               "`make -C psgd_tf_amd/csrc`)" % (LIB_PATH, lib.psgd_abi_version(), PSGD_ABI_VERSION))
        _lib = lib
    return _lib


def check(code, what):
    if code != PSGD_OK:
        msg = load().psgd_error_string(int(code)).decode()
        raise PsgdHipError("%s failed: %s (code %d)" % (what, msg, code))


def ws_region(which, stage, N, r):
    off, cnt = _i64(0), _i64(0)
    check(load().psgd_uvd_ws_region(which, stage, N, r, ctypes.byref(off), ctypes.byref(cnt)), "psgd_uvd_ws_region")
    return off.value, cnt.value


def splu_ws_region(which, stage, N, r):
    off, cnt = _i64(0), _i64(0)
    check(load().psgd_splu_ws_region(which, stage, N, r, ctypes.byref(off), ctypes.byref(cnt)), "psgd_splu_ws_region")
    return off.value, cnt.value


class WorkspaceCache:
    """Device workspaces keyed by (device, problem shape, stream), least-recently-used eviction.

    The C ABI never allocates: the caller owns the scratch memory.  A model has a fixed set of shapes, so this
    cache normally never evicts; the bounds keep a program that sweeps many shapes from accumulating
    workspaces (they are the size of a few operand copies each).  Eviction only drops the reference: work
    already queued on the stream keeps using the block, and torch's caching allocator reuses it in the order of the
    stream it was allocated on -- which is the stream in the key, the only one that ever uses the block."""

    def __init__(self, max_entries=32, max_bytes=32 << 30):
        import collections
        self._d = collections.OrderedDict()
        self.max_entries, self.max_bytes = max_entries, max_bytes

    def get(self, key, make):
        ws = self._d.get(key)
        if ws is not None:
            self._d.move_to_end(key)
            return ws
        ws = make()
        self._d[key] = ws
        total = sum(int(t.numel()) for t in self._d.values())
        while len(self._d) > 1 and (len(self._d) > self.max_entries or total > self.max_bytes):
            _, old = self._d.popitem(last=False)
            total -= int(old.numel())
        return ws

    def put(self, key, ws):
        """Install a block the caller allocated (placement.UVdArena puts its workspace region where the sweeps want it)."""
        self._d[key] = ws
        self._d.move_to_end(key)

    def touch(self, key):
        """Mark `key` as just used (callers that keep their own handle to a block call this instead of get())."""
        if key in self._d:
            self._d.move_to_end(key)

    def __contains__(self, key):
        return key in self._d

    def __getitem__(self, key):
        return self._d[key]

    def __len__(self):
        return len(self._d)
