"""Host-side logic of the drop-in module on CPU: dispatch table, index logic, hyper-parameter
holders, the dense plumbing path (config 1: hello_psgd) and the refusal to run the hot path
without the HIP device."""
import math

import numpy as np
import pytest
import torch

import preconditioned_stochastic_gradient_descent as psgd
from oracle import psgd_oracle as orc
from psgd_tf_amd import _lib, kron
from tests.test_oracle_kat import DISPATCH
from tests.uvd_cases import rel_err


def test_module_surface_matches_reference_names():
    # names imported by the reference's drivers (hello_psgd.py:25-26, mnist_with_lenet5.py:51,53,
    # rnn_xor_UVd_preconditioner.py:37, psgd.py:540,554,619)
    for name in ("update_precond_dense", "precond_grad_dense", "update_precond_kron", "precond_grad_kron",
                 "IpUVtmatvec", "update_precond_UVd_math_", "precond_grad_UVd_math", "UVd", "dtype", "_tiny"):
        assert hasattr(psgd, name), name
    assert psgd.dtype == torch.float32 and psgd._tiny == float(np.finfo(np.float32).tiny)
    import inspect
    sig = inspect.signature(psgd.UVd.__init__)
    assert list(sig.parameters)[1:9] == ["params_with_grad", "rank_of_modification", "preconditioner_init_scale",
                                         "lr_params", "lr_preconditioner", "grad_clip_max_norm",
                                         "preconditioner_update_probability", "exact_hessian_vector_product"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["rank_of_modification"], d["preconditioner_init_scale"], d["lr_params"], d["lr_preconditioner"],
            d["grad_clip_max_norm"], d["preconditioner_update_probability"], d["exact_hessian_vector_product"]) == \
        (10, 1.0, 0.01, 0.01, None, 1.0, True)                                      # psgd.py:663-666
    assert inspect.signature(psgd.update_precond_kron).parameters["step"].default == 0.01   # psgd.py:72
    assert inspect.signature(psgd.update_precond_dense).parameters["step"].default == 0.01  # psgd.py:26


@pytest.mark.parametrize("sl,sr,fmt", DISPATCH)
def test_dispatch_table_is_bit_exact_with_oracle(sl, sr, fmt):
    assert kron.kron_format(sl, sr) == fmt == orc.kron_format(sl, sr)


def test_param_index_logic():
    shapes = [(2, 30), (30, 30), (30,), (30, 1), (1,)]
    params = [torch.zeros(s) for s in shapes]
    sizes, cum = psgd.uvd_param_index(params)
    osizes, ocum = orc.uvd_param_index(shapes)
    assert sizes == osizes == [60, 900, 30, 30, 1] and cum == list(ocum) == [60, 960, 990, 1020, 1021]


def test_flatten_order_is_nest_flatten():
    from psgd_tf_amd.preconditioned_stochastic_gradient_descent import _flatten_params
    a, b, c, d = (torch.full((1,), float(i)) for i in range(4))
    flat = _flatten_params([a, [b, (c,)], {"z": d, "y": a}])
    assert [float(t) for t in flat] == [0.0, 1.0, 2.0, 0.0, 3.0]      # dict keys sorted, depth first
    assert _flatten_params(a) == [a]                                    # psgd.py:668


def test_hyper_parameters_assign_like_tf_variables():
    h = psgd.UVd.__init__.__globals__["_Hyper"](0.01)
    assert float(h) == 0.01
    h.assign(0.5)                                                      # psgd.py:660-661 (Note 4)
    assert float(h) == 0.5 and h.numpy() == 0.5
    flag = psgd.UVd.__init__.__globals__["_Hyper"](True)
    flag.assign(False)                                                 # rnn_xor_UVd_preconditioner.py:69
    assert not bool(flag)


def test_dense_plumbing_matches_oracle_and_kat_r():
    Q = 0.1 * torch.eye(2, dtype=torch.float64)
    vs = [torch.tensor(1.0, dtype=torch.float64), torch.tensor(0.0, dtype=torch.float64)]
    hvs = [torch.tensor(802.0, dtype=torch.float64), torch.tensor(400.0, dtype=torch.float64)]
    Qn = psgd.update_precond_dense(Q, vs, hvs, step=0.2)
    assert np.allclose(Qn.numpy(), [[0.08, -0.0101326], [0.0, 0.09494634]], rtol=2e-6, atol=1e-9)
    pg = psgd.precond_grad_dense(Qn, [torch.tensor(-4.0, dtype=torch.float64), torch.tensor(0.0, dtype=torch.float64)])
    assert pg[0].shape == () and np.allclose([float(pg[0]), float(pg[1])], [-0.0256, 0.00324243], rtol=2e-6)
    rng = np.random.default_rng(0)
    Qa = np.triu(rng.standard_normal((7, 7)) * 0.1, 1) + np.eye(7)
    dxs = [rng.standard_normal((2, 2)), rng.standard_normal(3)]
    dgs = [rng.standard_normal((2, 2)), rng.standard_normal(3)]
    got = psgd.update_precond_dense(torch.from_numpy(Qa), [torch.from_numpy(x) for x in dxs],
                                    [torch.from_numpy(x) for x in dgs], step=0.05)
    assert rel_err(got.numpy(), orc.update_precond_dense(Qa, dxs, dgs, 0.05)) < 1e-12
    pgs = psgd.precond_grad_dense(got, [torch.from_numpy(x) for x in dgs])
    refs = orc.precond_grad_dense(got.numpy(), dgs)
    assert all(a.shape == b.shape and rel_err(a.numpy(), b) < 1e-12 for a, b in zip(pgs, refs))


def test_hello_psgd_converges():
    """Harness row H of SURVEY 8a: f falls from f0 = 4 toward 0 over 500 iterations."""
    from examples.hello_psgd import run
    f, xs, Q = run(num_iter=500, seed=0)
    assert f[0] == pytest.approx(4.0) and f[-1] < 1e-8 and abs(xs[0] - 1) < 1e-3 and abs(xs[1] - 1) < 1e-3
    f1, _, Q1 = run(num_iter=1, first_v=(1.0, 0.0))
    assert np.allclose(Q1.numpy(), [[0.08, -0.0101326], [0.0, 0.09494634]], rtol=1e-5, atol=1e-8)


def test_hot_path_refuses_cpu_tensors():
    U, V = torch.zeros(10, 2), torch.zeros(10, 2)
    d, g = torch.ones(10, 1), torch.ones(10, 1)
    with pytest.raises(_lib.PsgdHipError, match="no CPU fallback"):
        psgd.precond_grad_UVd_math(U, V, d, g)
    with pytest.raises(_lib.PsgdHipError, match="no CPU fallback"):
        psgd.update_precond_UVd_math_(U, V, d, g, g, 0.01, 1e-38)
    with pytest.raises(_lib.PsgdHipError, match="no CPU fallback"):
        psgd.update_precond_kron(torch.eye(3), torch.eye(4), torch.ones(3, 4), torch.ones(3, 4))
    with pytest.raises(ValueError):
        psgd.update_precond_kron(torch.eye(3), torch.eye(4), torch.ones(12), torch.ones(12))   # rank-2 only (psgd.py:67-71)


def test_uvd_constants():
    assert torch.finfo(torch.float32).eps ** 0.5 == pytest.approx(2.0 ** -11.5)    # psgd.py:683
    assert math.isinf(float(psgd.UVd.__init__.__globals__["_Hyper"](math.inf)))     # psgd.py:675-676


def test_workspace_cache_is_bounded():
    """The Python boundary caches one device workspace per problem shape; a program that sweeps shapes must not
    accumulate them (LRU by entry count and by bytes; the newest entry always stays)."""
    import torch
    from psgd_tf_amd import _lib
    c = _lib.WorkspaceCache(max_entries=3, max_bytes=100)
    for i in range(5):
        c.get(i, lambda: torch.empty(10, dtype=torch.uint8))
    assert len(c) == 3 and 4 in c and 2 in c and 0 not in c
    hit = c.get(3, lambda: 1 / 0)                     # a hit does not rebuild and refreshes the entry
    assert hit.numel() == 10
    c.get(9, lambda: torch.empty(95, dtype=torch.uint8))
    assert 9 in c and len(c) == 1                      # byte bound: everything older had to go


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch
    it.  Static check over the product package, the root shim, tools/ and examples/, plus bench.py outside cpu_baseline."""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "preconditioned_stochastic_gradient_descent.py")]
    for sub in ("psgd_tf_amd", "tools", "examples"):
        for dp, _, fns in os.walk(os.path.join(root, sub)):
            files += [os.path.join(dp, f) for f in fns if f.endswith(".py")]

    def oracle_imports(tree):
        out = []
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                out += [(node.lineno, a.name) for a in node.names if a.name.split(".")[0] == "oracle"]
            elif isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "oracle":
                out.append((node.lineno, node.module))
        return out

    for f in files:
        assert oracle_imports(ast.parse(open(f).read())) == [], f
    # bench.py: the only import sits inside the cpu_baseline function
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    inside = []
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and "cpu_baseline" in node.name:
            inside += oracle_imports(node)
    assert sorted(oracle_imports(tree)) == sorted(inside) and inside
    # __graft_entry__: only smoke()
    tree = ast.parse(open(os.path.join(root, "__graft_entry__.py")).read())
    inside = []
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "smoke":
            inside += oracle_imports(node)
    assert sorted(oracle_imports(tree)) == sorted(inside)


def test_committed_pmc_traffic_file_is_what_bench_reads():
    """bench.py fills roofline.traffic from profiles/pmc_traffic.json (key k_update_s2 with the row count and rank of the
    headline workload); a raw per-kernel table copied over it would silently turn the field into null."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))["k_update_s2"]
    assert rec["rows"] == 100_000_000 and rec["r"] == 20
    alg = 4 * (3 * 20 + 5) * rec["rows"]                      # bytes the fused sweep moves: 260 B/row
    assert 0.98 * alg < rec["hbm_bytes_per_launch"] < 1.10 * alg


@pytest.mark.parametrize("sl,sr,sx", [((5, 5), (3, 3), (3, 5)),      # wrong-orientation data matrix
                                      ((4, 4), (3, 3), (5, 3)),      # left factor smaller than M
                                      ((5, 5), (2, 3), (5, 4)),      # (dense, normalization) with a short right factor
                                      ((1, 5), (3, 3), (4, 3)),      # (scaling, dense): 1 x M factor of the wrong length
                                      ((2, 5), (1, 3), (5, 2))])
def test_kron_factor_data_shape_mismatch_raises(sl, sr, sx):
    """The reference's matmuls raise on mismatched shapes; the kernels take raw pointers, so the host checks (and does so
    before anything device-side: CPU tensors are enough to see it)."""
    import preconditioned_stochastic_gradient_descent as psgd
    Ql, Qr, X = torch.ones(sl), torch.ones(sr), torch.ones(sx)
    with pytest.raises(ValueError, match="do not match"):
        psgd.precond_grad_kron(Ql, Qr, X)
    with pytest.raises(ValueError, match="do not match"):
        psgd.update_precond_kron(Ql, Qr, X, X, 0.01)


def test_kron_dX_dG_shape_mismatch_raises():
    import preconditioned_stochastic_gradient_descent as psgd
    with pytest.raises(ValueError, match="share one shape"):
        psgd.update_precond_kron(torch.eye(4), torch.eye(3), torch.ones(4, 3), torch.ones(3, 4), 0.01)


def test_factor_tag_identity_rules():
    """kron._FactorTag decides whether prepared, factor-only data (Grams, bf16 copies) may be reused: only for the very
    tensor objects it was made from, unmodified."""
    a, b = torch.eye(4), torch.eye(3)
    tag = kron._FactorTag((a, b))
    assert tag.matches((a, b))
    assert not tag.matches((a.clone(), b))                 # equal values, other object
    assert not tag.matches((b, a)) and not tag.matches((a,))
    a.mul_(2.0)                                            # in place: version counter moves
    assert not tag.matches((a, b))
    tag = kron._FactorTag((a, b))
    view = a.view(4, 4)                                    # another object on the same storage
    assert not tag.matches((view, b))


@pytest.mark.parametrize("M,N", [(5, 12), (13, 7), (1, 3)])
def test_zero_padding_of_bf16_kron_shapes_is_exact_in_the_oracle(M, N):
    """The bf16 kernels need M, N multiples of 8; other shapes run as blockdiag(Q, tiny I) with zero-padded data
    (kron._padded_bf16_problem).  Claim: every product and solve of psgd.py:156-192 is then block diagonal, so the leading
    block of each result is the unpadded result.  Checked here with the fp64 oracle on the padded problem."""
    rng = np.random.default_rng(M * 31 + N)
    tri = lambda n: np.triu(rng.standard_normal((n, n)) * 0.2, 1) + np.diag(np.exp(0.3 * rng.standard_normal(n)))
    Ql, Qr = tri(M), tri(N)
    dX, dG, G = (rng.standard_normal((M, N)) for _ in range(3))
    t = lambda a: torch.from_numpy(a)
    Qlp, Qrp, (dXp, dGp, Gp) = kron._padded_bf16_problem(t(Ql), t(Qr), (t(dX), t(dG), t(G)))
    assert Qlp.shape[0] % 8 == 0 and Qrp.shape[0] % 8 == 0 and tuple(Gp.shape) == (Qlp.shape[0], Qrp.shape[0])
    a, b = orc.update_precond_kron(Ql, Qr, dX, dG, 0.01)
    ap, bp = orc.update_precond_kron(Qlp.numpy(), Qrp.numpy(), dXp.numpy(), dGp.numpy(), 0.01)
    assert rel_err(ap[:M, :M], a) < 1e-12 and rel_err(bp[:N, :N], b) < 1e-12
    out = orc.precond_grad_kron(Ql, Qr, G)
    outp = orc.precond_grad_kron(Qlp.numpy(), Qrp.numpy(), Gp.numpy())
    assert rel_err(outp[:M, :N], out) < 1e-12 and np.abs(outp[M:]).max(initial=0.0) == 0.0 and np.abs(outp[:, N:]).max(initial=0.0) == 0.0


def test_wide_rank_chunking():
    from psgd_tf_amd import uvd_wide
    for r in (33, 36, 40, 48, 50, 64, 65, 70, 96, 100, 128, 257):
        c, rc, views = uvd_wide._chunks(r)
        cmin = -(-r // 32)
        assert rc <= 32 and c * rc >= r and (c - 1) * rc < r and c in (cmin, cmin + 1)
        if views:                                   # column views need an even split; their widths keep every view aligned
            assert c * rc == r
            lv = 4 if rc % 4 == 0 else (2 if rc % 2 == 0 else 1)
            assert r % lv == 0 and all((k * rc) % lv == 0 for k in range(c))
        else:
            assert c == cmin and r % cmin != 0 and (r % (cmin + 1) != 0 or r // (cmin + 1) > 32)
    assert uvd_wide._chunks(40) == (2, 20, True) and uvd_wide._chunks(33) == (3, 11, True) and uvd_wide._chunks(70) == (3, 24, False)


def test_layer_streams_block_on_cpu_tensors_raises_the_entry_points_error():
    """kron.layer_streams only forks device calls: a CPU tensor inside the block reaches the entry point unforked (which refuses it:
    no CPU fallback) instead of recursing, and the block's switch is cleared on the way out."""
    import torch
    import preconditioned_stochastic_gradient_descent as psgd
    from psgd_tf_amd import _lib, kron
    with pytest.raises(_lib.PsgdHipError):
        with kron.layer_streams():
            psgd.precond_grad_kron(torch.eye(3), torch.eye(2), torch.ones(3, 2))
    assert kron._layer_ctx is None
    with pytest.raises(ValueError):
        kron.layer_streams(0)
