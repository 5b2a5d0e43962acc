"""The C-ABI library: builds for gfx950 without a GPU, loads, and exports exactly the symbols
that include/psgd_hip.h declares (and _lib.py binds).  No compute calls here -- only entry points
whose argument checks return before any HIP call."""
import ctypes
import os
import re

import pytest

from psgd_tf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "psgd_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psgd_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build_extension()
    return _lib.load()


def test_header_symbols_are_exported_and_bound(lib):
    declared = _declared_symbols()
    assert len(declared) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "declared in psgd_hip.h but not exported: " + name
        assert name in _lib.SIGNATURES, "declared in psgd_hip.h but not bound in _lib.py: " + name
    assert sorted(_lib.SIGNATURES) == declared


def test_abi_version_and_error_strings(lib):
    assert lib.psgd_abi_version() == _lib.PSGD_ABI_VERSION == 7
    assert lib.psgd_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -6):
        assert len(lib.psgd_error_string(code)) > 3


def test_workspace_queries_and_argument_checks(lib):
    # psgd.py:690 keeps an N-vector temporary per call (nablaD, :581): workspace grows by 4 N bytes
    a = lib.psgd_uvd_workspace_bytes(1_000_000, 10)
    b = lib.psgd_uvd_workspace_bytes(2_000_000, 10)
    assert a > 0 and b - a == 4_000_000 and a % 256 == 0
    assert lib.psgd_uvd_workspace_bytes(100, 0) == -2 and lib.psgd_uvd_workspace_bytes(100, 33) == -2   # PSGD_ERR_RANK
    assert lib.psgd_uvd_workspace_bytes(0, 4) == -1                                                       # PSGD_ERR_BAD_ARG
    off, cnt = _lib.ws_region(_lib.PSGD_WS_SUMS_F64, 1, 1000, 20)
    off2, cnt2 = _lib.ws_region(_lib.PSGD_WS_SUMS_F64, 2, 1000, 20)
    assert cnt == cnt2 == 20 and off2 - off == 160
    _, gram = _lib.ws_region(_lib.PSGD_WS_SUMS_F64, 11, 1000, 20)
    assert gram == 6 * 256                      # [U V t w] = 42 columns -> 3 MFMA blocks -> 6 block pairs
    _, gram10 = _lib.ws_region(_lib.PSGD_WS_SUMS_F64, 11, 1000, 10)
    assert gram10 == 3 * 256
    _, m = _lib.ws_region(_lib.PSGD_WS_MAX_F32, 10, 1000, 20)
    assert m == 2
    # null pointers are rejected before anything touches the device
    assert lib.psgd_uvd_apply_f32(None, None, None, None, None, 100, 4, None, 0, None) == -1
    assert lib.psgd_uvd_update_f32(None, None, None, None, None, 100, 4, 0.01, 1e-38, 0, 1, None, 0, None) == -1
    assert lib.psgd_kron_dd_apply_f32(None, None, None, None, 4, 4, None, 0, None) == -1
    assert lib.psgd_kron_dd_workspace_bytes(0, 4) == -1
    assert lib.psgd_kron_dd_workspace_bytes(257, 120) > 4 * (2 * 257 * 257 + 2 * 120 * 120 + 4 * 257 * 120)


def test_kron_workspace_layout_rules(lib):
    """Workspace sizes are pure functions of the shape (a workspace sized once stays valid whatever the tuning keys): the plane
    buffers appear from the planes threshold on, the buffers of the solves through explicit inverses from 2048 x 2048 on --
    in the fp32 workspace and in the bf16-operand update's alike -- and tuning keys do not move the sizes."""
    ws = lib.psgd_kron_dd_workspace_bytes
    fp32_only = lambda m, n: 4 * (4 * m * m + 2 * n * n + 4 * m * n)          # a lower bound of the fp32 buffers
    assert ws(300, 300) < 3 * fp32_only(300, 300)                             # no planes at 300^2
    assert ws(1024, 1024) > fp32_only(1024, 1024) + 16 * 1024 * 1024 * 6      # planes of the large paths (6 B per element)
    per = lambda n: ws(n, n) / float(n * n)
    assert per(2048) > per(1920) + 4 * 4                                     # inverse route: more bytes per element from 2048^2 on
    upd = lib.psgd_kron_dd_update_workspace_bytes_bf16
    assert upd(2048, 2048) / float(2048 * 2048) > upd(1920, 1920) / float(1920 * 1920) + 4 * 4
    before = (ws(2048, 2048), upd(2048, 2048), ws(1024, 1024))
    try:
        for key, val in ((11, 0), (12, 0), (4, 0)):
            assert lib.psgd_kron_set_tuning(key, val) == 0
        assert (ws(2048, 2048), upd(2048, 2048), ws(1024, 1024)) == before
    finally:
        for key, val in ((11, 1), (12, 2), (4, 1)):
            lib.psgd_kron_set_tuning(key, val)
    assert lib.psgd_kron_set_tuning(999, 0) == -1                             # unknown key: PSGD_ERR_BAD_ARG
    sp = lib.psgd_kron_sparse_workspace_bytes
    assert sp(1, 30000, 1000) > sp(1, 30000, 500) and sp(2, 30000, 1000) < sp(1, 30000, 1000)      # (norm, dense) vs (norm, scaling)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PsgdHipError, match="no CPU fallback|missing"):
        _lib.load()
