"""GPU parity: Kronecker-product preconditioner (HIP dense(x)dense through the C ABI, the sparse
formats through the same dispatcher) vs the CPU oracle on identical seeded inputs.

Tolerance: 1e-5 relative (norm-wise) against the fp64 oracle for the preconditioned gradient
and for the updated factors; the update increment (~step of the factor) within 2e-3.
"""
import numpy as np
import pytest
import torch

from oracle import psgd_oracle as orc
from tests.uvd_cases import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5
INCR_TOL = 2e-3

# LeNet5 affine shapes (mnist_with_lenet5.py:12-16), LSTM shapes (lstm_with_xor_problem.py),
# the 1x1 dense factor of the NMT demo (:124), tile edges of the 64x64 GEMM blocks and 32-wide
# triangular-solve blocks, and one larger case.
DD_SHAPES = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10), (63, 120), (31, 1), (1, 1), (1, 7), (3, 3),
             (2, 2), (64, 64), (65, 33), (32, 97), (128, 200), (300, 500), (512, 384), (700, 600), (1100, 530)]


def _tri_factor(rng, n, off=0.05):
    return np.triu(rng.standard_normal((n, n)) * off, 1) + np.diag(np.exp(0.3 * rng.standard_normal(n)))


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


@pytest.fixture(scope="module")
def psgd(hip_lib):
    import preconditioned_stochastic_gradient_descent as m
    return m


@pytest.fixture
def stage_kernels(hip_lib):
    """(Round 4's opt-in fused strip kernels are gone: single calls on small layers always run the stage kernels -- Grams + products,
    batch-of-one update -- which the tests taking this fixture vary.)"""
    yield


@pytest.mark.parametrize("M,N", DD_SHAPES)
def test_dense_dense_apply(psgd, M, N):
    rng = np.random.default_rng(M * 1000 + N)
    Ql, Qr = _tri_factor(rng, M), _tri_factor(rng, N)
    G = rng.standard_normal((M, N))
    Ql32, Qr32, G32 = (a.astype(np.float32) for a in (Ql, Qr, G))
    out = psgd.precond_grad_kron(_dev(Ql32), _dev(Qr32), _dev(G32))
    ref = orc.precond_grad_kron(Ql32.astype(np.float64), Qr32.astype(np.float64), G32.astype(np.float64))
    assert out.shape == (M, N)
    assert rel_err(out.cpu().numpy(), ref) < TOL


# shapes whose GEMM stages run on pre-split operand planes (tile multiples, zero-padded edges, both aspect ratios, skinny)
PLANE_SHAPES = [(1024, 1024), (1030, 1100), (1155, 1024), (1024, 1290), (512, 512), (600, 530), (260, 3100), (8200, 70)]


@pytest.mark.parametrize("M,N", DD_SHAPES + PLANE_SHAPES)
def test_dense_dense_update(psgd, M, N):
    rng = np.random.default_rng(M * 77 + N)
    Ql, Qr = _tri_factor(rng, M) * 3.0, _tri_factor(rng, N)      # rho != 1
    dX = rng.standard_normal((M, N))
    Hl = np.eye(M) + 0.1 * np.diag(rng.uniform(0, 5, M))
    Hr = np.eye(N) + 0.1 * np.diag(rng.uniform(0, 5, N))
    dG = Hl @ dX @ Hr
    a32 = [a.astype(np.float32) for a in (Ql, Qr, dX, dG)]
    dQl, dQr = _dev(a32[0]), _dev(a32[1])
    Ql_n, Qr_n = psgd.update_precond_kron(dQl, dQr, _dev(a32[2]), _dev(a32[3]), 0.01)
    a64 = [a.astype(np.float64) for a in a32]
    Ql_r, Qr_r = orc.update_precond_kron(*a64, 0.01)
    assert rel_err(Ql_n.cpu().numpy(), Ql_r) < TOL
    assert rel_err(Qr_n.cpu().numpy(), Qr_r) < TOL
    # pure: inputs untouched (psgd.py:179 returns new tensors)
    assert np.array_equal(dQl.cpu().numpy(), a32[0]) and np.array_equal(dQr.cpu().numpy(), a32[1])
    # increments relative to the balanced factors (psgd.py:169-170)
    rho = np.sqrt(np.max(np.diag(a64[0])) / np.max(np.diag(a64[1])))
    for got, ref, base in ((Ql_n, Ql_r, a64[0] / rho), (Qr_n, Qr_r, a64[1] * rho)):
        inc_ref = ref - base
        if np.linalg.norm(inc_ref) > 0:
            assert rel_err(got.cpu().numpy().astype(np.float64) - base, inc_ref) < INCR_TOL
    # KAT-TRI: upper-triangular with positive diagonal
    for Qn in (Ql_n, Qr_n):
        q = Qn.cpu().numpy()
        assert np.array_equal(q, np.triu(q)) and (np.diag(q) > 0).all()


def test_dense_dense_sequence_lenet(psgd):
    """Ten update steps from Ql = I, Qr = I on the LeNet5 W3 shape, then an apply (mnist_with_lenet5.py:51-53)."""
    M, N = 257, 120
    rng = np.random.default_rng(0)
    Ql, Qr = np.eye(M), np.eye(N)
    tQl, tQr = _dev(Ql), _dev(Qr)
    Hl = np.diag(np.exp(rng.uniform(-1, 1, M)))
    Hr = np.diag(np.exp(rng.uniform(-1, 1, N)))
    for _ in range(10):
        dX = rng.standard_normal((M, N)).astype(np.float32)
        dG = (Hl @ dX @ Hr).astype(np.float32)
        tQl, tQr = psgd.update_precond_kron(tQl, tQr, _dev(dX), _dev(dG), 0.01)
        Ql, Qr = orc.update_precond_kron(Ql, Qr, dX.astype(np.float64), dG.astype(np.float64), 0.01)
    assert rel_err(tQl.cpu().numpy(), Ql) < 5e-5 and rel_err(tQr.cpu().numpy(), Qr) < 5e-5
    G = rng.standard_normal((M, N)).astype(np.float32)
    out = psgd.precond_grad_kron(tQl, tQr, _dev(G))
    assert rel_err(out.cpu().numpy(), orc.precond_grad_kron(Ql, Qr, G.astype(np.float64))) < 5e-5


# every dispatch format of psgd.py:80-110 with shapes from demo_usage_of_all_preconditioners.py:68-78
FORMATS = {
    "dense_norm": ((5, 5), (2, 10)), "dense_scale": ((5, 5), (1, 20)), "norm_dense": ((2, 5), (10, 10)),
    "norm_scale": ((2, 5), (1, 50)), "scale_dense": ((1, 5), (20, 20)), "scale_norm": ((1, 5), (2, 50)),
}


def _factor_for(rng, shape, other_dim, last=0.0):
    m, n = shape
    if m == n:
        return _tri_factor(rng, m)
    if m == 2:
        q = np.stack([np.exp(0.2 * rng.standard_normal(n)), 0.1 * rng.standard_normal(n)])
        # psgd.py:205 documents the last entry of the stored column as unused (it starts at 0 and the update keeps it 0:
        # grad1_bias ends with 0, :238), but the arithmetic of :219, :232, :265 READS it: a caller's non-zero value takes part
        q[1, -1] = last
        return q
    return np.exp(0.2 * rng.standard_normal((1, n)))


@pytest.mark.parametrize("fmt,last", [(f, 0.0) for f in sorted(FORMATS)] + [(f, 0.07) for f in sorted(FORMATS) if "norm" in f])
def test_sparse_formats_through_dispatcher(psgd, fmt, last):
    sl, sr = FORMATS[fmt]
    assert orc.kron_format(sl, sr) == fmt
    rng = np.random.default_rng(len(fmt))
    M, N = sl[1], sr[1]
    Ql, Qr = _factor_for(rng, sl, M, last), _factor_for(rng, sr, N, last)
    dX, G = rng.standard_normal((M, N)), rng.standard_normal((M, N))
    dG = dX * np.exp(rng.uniform(-1, 1, (M, 1))) * np.exp(rng.uniform(-1, 1, (1, N)))
    a32 = [a.astype(np.float32) for a in (Ql, Qr, dX, dG, G)]
    a64 = [a.astype(np.float64) for a in a32]
    Ql_n, Qr_n = psgd.update_precond_kron(_dev(a32[0]), _dev(a32[1]), _dev(a32[2]), _dev(a32[3]), 0.01)
    Ql_r, Qr_r = orc.update_precond_kron(a64[0], a64[1], a64[2], a64[3], 0.01)
    assert Ql_n.shape == sl and Qr_n.shape == sr
    assert rel_err(Ql_n.cpu().numpy(), Ql_r) < TOL and rel_err(Qr_n.cpu().numpy(), Qr_r) < TOL
    out = psgd.precond_grad_kron(_dev(a32[0]), _dev(a32[1]), _dev(a32[4]))
    assert rel_err(out.cpu().numpy(), orc.precond_grad_kron(a64[0], a64[1], a64[4])) < TOL


def test_unknown_format_passthrough(psgd, capsys):
    Ql, Qr = torch.ones(2, 5, device="cuda"), torch.ones(2, 7, device="cuda")      # (norm, norm): unknown
    G = torch.randn(5, 7, device="cuda")
    a, b = psgd.update_precond_kron(Ql, Qr, G, G, 0.01)
    assert a is Ql and b is Qr                                                      # psgd.py:97-99
    assert psgd.precond_grad_kron(Ql, Qr, G) is G                                   # psgd.py:139-141
    assert "Unknown Kronecker product preconditioner" in capsys.readouterr().out


# ----------------------------------------------------------------------------- bf16-operand apply (config 5)
BF16_TOL = 2e-2      # 4 chained bf16-operand GEMMs with bf16 intermediates: ~2^-8 per rounding, see DESIGN.md 4.4


@pytest.mark.parametrize("M,N", [(128, 128), (256, 512), (512, 256), (136, 264), (1024, 1024), (8, 8), (200, 72),
                                 (4096, 2048), (2048, 4096),       # these two: 8x8 tile-patch scheduling path
                                 (133, 260), (7, 5), (1, 3), (1027, 515), (4100, 4093),    # not multiples of 8: padded
                                 (2000, 2400), (1900, 2300)])      # large, not multiples of 256: padded to the 256-tile kernels
def test_dense_dense_apply_bf16(psgd, M, N):
    rng = np.random.default_rng(M + 3 * N)
    Ql, Qr = _tri_factor(rng, M).astype(np.float32), _tri_factor(rng, N).astype(np.float32)
    G = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda().to(torch.bfloat16)
    out = psgd.precond_grad_kron(_dev(Ql), _dev(Qr), G)
    assert out.dtype == torch.bfloat16 and out.shape == (M, N)
    ref = orc.precond_grad_kron(Ql.astype(np.float64), Qr.astype(np.float64), G.float().cpu().numpy().astype(np.float64))
    assert rel_err(out.float().cpu().numpy(), ref) < BF16_TOL
    # and it agrees with the fp32 HIP path on the same (bf16-valued) gradient to the same tolerance
    out32 = psgd.precond_grad_kron(_dev(Ql), _dev(Qr), G.float())
    assert rel_err(out.float().cpu().numpy(), out32.cpu().numpy()) < BF16_TOL
    from psgd_tf_amd import _lib
    variants = [1]                                    # 128^2 register-staged everywhere
    if M * N >= 4096 * 2048:
        variants.append(2)                            # 128^2 LDS-DMA ring
    if M % 256 == 0 and N % 256 == 0:
        variants.append(3)                            # 256^2 8-phase kernel for every product it can take
        variants.append(4)                            # auto without the fused triangular pair
    outs = {}
    for variant in variants:
        _lib.load().psgd_kron_bf16_set_tuning(0, variant)
        try:
            outs[variant] = psgd.precond_grad_kron(_dev(Ql), _dev(Qr), G)
        finally:
            _lib.load().psgd_kron_bf16_set_tuning(0, 0)
        assert rel_err(outs[variant].float().cpu().numpy(), ref) < BF16_TOL, variant
    # every variant multiplies the same bf16 operands in the same k order with fp32 accumulation: bitwise equal
    for variant in variants[1:]:
        assert torch.equal(outs[variant], outs[1]), variant
    # the default may take the fused triangular pairs: K chunks summed in descending order and the Gram re-associated
    # into the chain ((G Qr') Qr), so single bf16 roundings fall elsewhere
    assert rel_err(out.float().cpu().numpy(), outs[1].float().cpu().numpy()) < 1e-2


@pytest.mark.parametrize("M,N", [(4096, 4096), (2048, 1024), (256, 256)])
def test_bf16_gemm_variants_bitwise_stable(psgd, M, N):
    """Race screen for the LDS-DMA kernels (counted vmcnt / raw barriers): repeated runs of the 256^2 kernel on
    every product of the chain must reproduce the register-staged 128^2 kernel bit for bit."""
    from psgd_tf_amd import _lib
    rng = np.random.default_rng(M + N)
    Ql, Qr = _dev(_tri_factor(rng, M).astype(np.float32)), _dev(_tri_factor(rng, N).astype(np.float32))
    G = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda().to(torch.bfloat16)
    lib = _lib.load()
    lib.psgd_kron_bf16_set_tuning(0, 1)
    try:
        want = psgd.precond_grad_kron(Ql, Qr, G)
        for variant in (3, 4):
            lib.psgd_kron_bf16_set_tuning(0, variant)
            for _ in range(6):
                assert torch.equal(psgd.precond_grad_kron(Ql, Qr, G), want), variant
    finally:
        lib.psgd_kron_bf16_set_tuning(0, 0)


@pytest.mark.parametrize("M,N", [(4096, 4096), (4096, 2048), (8192, 1024), (2048, 4096), (1024, 8192),
                                 (8192, 4096), (6144, 6144)])      # the last two: several column-group launches
def test_bf16_fused_triangular_pair(psgd, M, N):
    """The wavefront-scheduled fused launch of  T3 = Ql T2,  out = Ql' T3  (in-launch hand-offs between workgroups).
    Checked against the two separate products on a stream of DIFFERENT gradients, so a hand-off that read a stale T3
    tile (the workspace still holds the previous call's) cannot pass; repeated calls must agree bit for bit; no spin
    may time out."""
    from psgd_tf_amd import _lib, kron
    rng = np.random.default_rng(M + N)
    Ql, Qr = _dev(_tri_factor(rng, M, 0.01).astype(np.float32)), _dev(_tri_factor(rng, N, 0.01).astype(np.float32))
    Gs = [torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda().to(torch.bfloat16) for _ in range(3)]
    lib = _lib.load()
    try:
        lib.psgd_kron_bf16_set_tuning(0, 1)
        want = [psgd.precond_grad_kron(Ql, Qr, G) for G in Gs]
        lib.psgd_kron_bf16_set_tuning(0, 0)
        # two_pairs = 0: Gram first, then ONE fused pair (same association as the staged chain: <= 1 bf16 ulp apart);
        # two_pairs = 1 (default): (G Qr') Qr then Ql' (Ql .): one bf16 rounding sits elsewhere
        for two_pairs, tol in ((0, 2e-3), (1, 1e-2)):
            lib.psgd_kron_bf16_set_tuning(1, two_pairs)
            first = None
            for rep in range(3):
                for i in (0, 1, 2, 1, 0):
                    got = psgd.precond_grad_kron(Ql, Qr, Gs[i])
                    assert rel_err(got.float().cpu().numpy(), want[i].float().cpu().numpy()) < tol, (two_pairs, rep, i)
                    if i == 2:
                        if first is None:
                            first = got
                        assert torch.equal(got, first), (two_pairs, rep)
            assert kron.check_bf16_handoffs() == 0          # no hand-off ever ran into its wait bound
    finally:
        lib.psgd_kron_bf16_set_tuning(0, 0)
        lib.psgd_kron_bf16_set_tuning(1, 1)


def test_bf16_fused_pair_on_a_shared_device(psgd):
    """The fused pair's schedule assumes its workgroups resident; the launcher can only check that against the CU count,
    not against other streams.  Its RESULT must not depend on it.  (a) With a second stream keeping the CUs busy (large
    fp32 GEMMs) the calls give the idle-device result bit for bit -- late workgroups only delay their consumers.
    (b) When consumers do give up waiting (provoked here by a poll bound of 1: nearly every hand-off times out), they
    produce the missing tile themselves and start over: the result is still the idle-device result bit for bit -- finite
    and correct, not NaN -- and the recoveries are counted."""
    from psgd_tf_amd import _lib, kron
    lib = _lib.load()
    M = N = 4096
    rng = np.random.default_rng(7)
    Ql, Qr = _dev(_tri_factor(rng, M, 0.01).astype(np.float32)), _dev(_tri_factor(rng, N, 0.01).astype(np.float32))
    G = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda().to(torch.bfloat16)
    kron.HANDOFF_FALLBACK_AFTER = 1 << 30                  # (this test provokes recoveries on purpose: the automatic switch stays out of it)
    want = psgd.precond_grad_kron(Ql, Qr, G)
    torch.cuda.synchronize()
    assert torch.isfinite(want.float()).all()
    ref = orc.precond_grad_kron(*(t.float().cpu().numpy().astype(np.float64) for t in (Ql, Qr, G)))
    assert rel_err(want.float().cpu().numpy(), ref) < 2e-2
    base = kron.check_bf16_handoffs()
    # (a) contention
    side = torch.cuda.Stream()
    A = torch.randn(8192, 8192, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(40):                                 # ~0.3 s of GEMMs that fill every CU
            A = torch.mm(A, A) * 1e-4
    outs = [psgd.precond_grad_kron(Ql, Qr, G) for _ in range(30)]
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)
    base = kron.check_bf16_handoffs()
    # (b) consumers that give up: same bits, recoveries counted
    try:
        assert lib.psgd_kron_bf16_set_tuning(2, 0) == 0
        forced = [psgd.precond_grad_kron(Ql, Qr, G) for _ in range(5)]
        torch.cuda.synchronize()
    finally:
        lib.psgd_kron_bf16_set_tuning(2, 22)
    for o in forced:
        assert torch.isfinite(o.float()).all()
        assert torch.equal(o, want)
    assert kron.check_bf16_handoffs() > base                 # the bound of 1 poll was hit and recovered from
    # (c) the same under contention AND the tiny bound, with changing gradients (a stale tile cannot pass)
    Gs = [torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda().to(torch.bfloat16) for _ in range(3)]
    wants = [psgd.precond_grad_kron(Ql, Qr, g) for g in Gs]
    torch.cuda.synchronize()
    try:
        lib.psgd_kron_bf16_set_tuning(2, 0)
        with torch.cuda.stream(side):
            for _ in range(20):
                A = torch.mm(A, A) * 1e-4
        got = [psgd.precond_grad_kron(Ql, Qr, g) for g in Gs]
        torch.cuda.synchronize()
    finally:
        lib.psgd_kron_bf16_set_tuning(2, 22)
    for o, w_ in zip(got, wants):
        assert torch.equal(o, w_)
    for key, (m, n) in kron._bf16_apply_shapes.items():          # clear the counters for the tests that follow
        if key in kron._kron_ws_bf16:
            lib.psgd_kron_bf16_handoff_reset(kron._kron_ws_bf16[key].data_ptr(), m, n, None)
    torch.cuda.synchronize()
    assert kron.check_bf16_handoffs() == 0
    kron.HANDOFF_FALLBACK_AFTER = 3
    kron.reset_bf16_handoff_fallback()
    assert not kron.bf16_handoff_fallback_active()


def test_bf16_handoff_fallback_switches_by_itself(psgd):
    """Round 6: the library watches the recovery counter of a bf16 apply workspace without synchronising (an asynchronous copy of
    the word every HANDOFF_CHECK_EVERY-th call) and, after HANDOFF_FALLBACK_AFTER recoveries, switches to the kernels without
    in-launch hand-offs by itself -- a shared device then stops paying the wait bound on every call.  Provoked with a poll bound
    of 1; the results before and after the switch agree with the fp64 oracle, and reset_bf16_handoff_fallback() goes back."""
    import warnings
    from psgd_tf_amd import _lib, kron
    lib = _lib.load()
    M, N = 2048, 2304
    rng = np.random.default_rng(17)
    Ql, Qr = _dev(_tri_factor(rng, M, 0.01).astype(np.float32)), _dev(_tri_factor(rng, N, 0.01).astype(np.float32))
    G = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda().to(torch.bfloat16)
    ref = orc.precond_grad_kron(*(t.float().cpu().numpy().astype(np.float64) for t in (Ql, Qr, G)))
    kron.reset_bf16_handoff_fallback()
    want = psgd.precond_grad_kron(Ql, Qr, G)
    torch.cuda.synchronize()
    keep = kron.HANDOFF_CHECK_EVERY
    try:
        kron.HANDOFF_CHECK_EVERY = 2
        lib.psgd_kron_bf16_set_tuning(2, 0)                                   # nearly every hand-off gives up and recovers
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for i in range(40):
                out = psgd.precond_grad_kron(Ql, Qr, G)
                torch.cuda.synchronize()
                if kron.bf16_handoff_fallback_active():
                    break
        assert kron.bf16_handoff_fallback_active(), "no switch after 40 calls with recoveries"
        assert any("hand-offs" in str(w_.message) for w_ in caught)
        assert torch.equal(out, want)                                          # (the call that noticed still ran the fused pair)
        safe = psgd.precond_grad_kron(Ql, Qr, G)                               # ... this one runs without in-launch hand-offs
        torch.cuda.synchronize()
        assert torch.isfinite(safe.float()).all() and rel_err(safe.float().cpu().numpy(), ref) < 2e-2
    finally:
        kron.HANDOFF_CHECK_EVERY = keep
        lib.psgd_kron_bf16_set_tuning(2, 22)
        kron.reset_bf16_handoff_fallback()
        for key, (m, n) in kron._bf16_apply_shapes.items():
            if key in kron._kron_ws_bf16:
                lib.psgd_kron_bf16_handoff_reset(kron._kron_ws_bf16[key].data_ptr(), m, n, None)
        torch.cuda.synchronize()
    assert not kron.bf16_handoff_fallback_active()
    assert torch.equal(psgd.precond_grad_kron(Ql, Qr, G), want)
    assert torch.equal(psgd.precond_grad_kron(Ql, Qr, G), want)


@pytest.mark.parametrize("M,N", [(512, 384), (1100, 530), (640, 256), (1024, 1536), (48, 1040)])
def test_triangular_solve_kernel_variants_agree(psgd, hip_lib, M, N):
    """The strip solve has three bodies: register-resident fully unrolled (full 512-column strips with 16 valid vectors
    per workgroup), register-resident general, and the LDS-resident one (psgd_kron_set_tuning(2, 1)).  Same block
    algorithm, different fp32 summation orders: the updated factors agree to the parity tolerance, and with the oracle."""
    rng = np.random.default_rng(M + 7 * N)
    Ql, Qr = (_tri_factor(rng, M) * 2.0).astype(np.float32), _tri_factor(rng, N).astype(np.float32)
    dX = rng.standard_normal((M, N)).astype(np.float32)
    dG = (dX * np.exp(rng.uniform(-1, 1, (1, N)))).astype(np.float32)
    outs = []
    try:
        for lds in (0, 1):
            hip_lib.psgd_kron_set_tuning(2, lds)
            outs.append(psgd.update_precond_kron(_dev(Ql), _dev(Qr), _dev(dX), _dev(dG), 0.01))
    finally:
        hip_lib.psgd_kron_set_tuning(2, 0)
    ref = orc.update_precond_kron(*(a.astype(np.float64) for a in (Ql, Qr, dX, dG)), 0.01)
    for k in range(2):
        assert rel_err(outs[0][k].cpu().numpy(), outs[1][k].cpu().numpy()) < TOL
        assert rel_err(outs[0][k].cpu().numpy(), ref[k]) < TOL
        assert rel_err(outs[1][k].cpu().numpy(), ref[k]) < TOL


@pytest.mark.parametrize("M,N,bf16", [(600, 530, False), (1024, 1024, False), (1100, 520, False), (70, 2100, False),
                                      (1024, 1024, True), (1536, 640, True)])
def test_update_chains_on_two_streams_change_nothing(psgd, hip_lib, M, N, bf16):
    """psgd_kron_set_tuning(9, .): the products of psgd.py:173 run on a side stream next to the solves of :174 (event
    fork/join inside the call).  Same kernels, same data: bitwise equal to the serial order, call after call (the side
    stream and its events are reused), and the next call on the caller's stream sees the finished result."""
    rng = np.random.default_rng(M + 3 * N)
    Ql, Qr = _dev(_tri_factor(rng, M) * 1.5), _dev(_tri_factor(rng, N))
    dX = _dev(rng.standard_normal((M, N)))
    dG = _dev(rng.standard_normal((M, N)) * 2.0)
    if bf16:
        dX, dG = dX.to(torch.bfloat16), dG.to(torch.bfloat16)
    outs = {}
    try:
        for key in (0, 1):
            hip_lib.psgd_kron_set_tuning(9, key)
            a, b = Ql, Qr
            for _ in range(3):                                # factors feed back: every call waits for the one before
                a, b = psgd.update_precond_kron(a, b, dX, dG, 0.01)
            outs[key] = (a.clone(), b.clone(), psgd.precond_grad_kron(a, b, dG).clone())
    finally:
        hip_lib.psgd_kron_set_tuning(9, 1)
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)


def test_updates_on_two_caller_streams_keep_their_own_side_chains(psgd):
    """Two caller streams issue updates back to back without synchronising: each has its own workspace, side stream and
    events, so the results are those of the same calls made alone."""
    rng = np.random.default_rng(123)
    probs = []
    for M, N in ((1024, 768), (640, 1100)):
        probs.append((_dev(_tri_factor(rng, M) * 1.5), _dev(_tri_factor(rng, N)), _dev(rng.standard_normal((M, N))),
                      _dev(rng.standard_normal((M, N)) * 2.0)))
    want = []
    for Ql, Qr, dX, dG in probs:
        a, b = Ql, Qr
        for _ in range(4):
            a, b = psgd.update_precond_kron(a, b, dX, dG, 0.01)
        want.append((a.clone(), b.clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    state = [(p[0], p[1]) for p in probs]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for _ in range(4):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                state[i] = psgd.update_precond_kron(state[i][0], state[i][1], probs[i][2], probs[i][3], 0.01)
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(state[i][0], want[i][0]) and torch.equal(state[i][1], want[i][1])


@pytest.mark.parametrize("warm,big", [(True, False), (False, False), (True, True), (False, True), (True, 4096)])
def test_update_with_forked_chains_is_capturable(psgd, warm, big):
    """The fork/join is made of events only, so an update can be captured into a graph on the caller's stream; the replay
    gives the eager result.  warm = False: the capture stream has never made an update call, so the library has no side
    stream for it yet and does not create one during the capture: that call stays serial (and its workspace is allocated
    from the graph's pool).  big: a shape whose solves run through explicit inverses (two inversions side by side)."""
    M = N = 1024 if warm else 896
    if big:
        M, N = (2304, 2048) if warm else (2048, 2176)
    if big == 4096:                                           # (round 6: three streams, the products of :173 on the third one)
        M = N = 4096
    rng = np.random.default_rng(77)
    Ql, Qr = _dev(_tri_factor(rng, M) * 1.5), _dev(_tri_factor(rng, N))
    dX, dG = _dev(rng.standard_normal((M, N))), _dev(rng.standard_normal((M, N)) * 2.0)
    want = [t.clone() for t in psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2 if warm else 0):                     # workspace and side stream of this stream exist before the capture
            psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        got = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
    for t in got:
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for x, y in zip(got, want):
        assert torch.equal(x, y)


@pytest.mark.parametrize("M,N", [(26, 6), (257, 120), (85, 10), (1, 1), (2, 3), (300, 300), (512, 40), (400, 300), (512, 512)])
def test_small_update_routes_agree(psgd, hip_lib, stage_kernels, M, N):
    """psgd_kron_set_tuning(7, .): the small-layer update as one launch per stage of each chain on the large-layer path
    (0), with product and solve stages sharing launches in the batched form (bit 0), and with single calls routed through
    a batch of one (bit 1, default 3).  Same block algorithms: batched results bitwise equal with and without shared
    launches; the single-call routes agree to rounding and with the oracle."""
    rng = np.random.default_rng(5 * M + N)
    Ql, Qr = _dev(_tri_factor(rng, M) * 1.5), _dev(_tri_factor(rng, N))
    dX = _dev(rng.standard_normal((M, N)))
    dG = _dev(rng.standard_normal((M, N)) * 2.0)
    outs = {}
    try:
        for key in (0, 1, 3):
            hip_lib.psgd_kron_set_tuning(7, key)
            u = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
            ub = psgd.update_precond_kron_batched([Ql], [Qr], [dX], [dG], 0.01)[0]
            outs[key] = [t.clone() for t in (u[0], u[1], ub[0], ub[1])]
    finally:
        hip_lib.psgd_kron_set_tuning(7, 3)
    ref = orc.update_precond_kron(*(t.cpu().numpy().astype(np.float64) for t in (Ql, Qr, dX, dG)), 0.01)
    for k in range(2):
        assert torch.equal(outs[0][2 + k], outs[1][2 + k])            # batched: shared launches change nothing
        assert torch.equal(outs[3][k], outs[3][2 + k])                # a single call IS a batch of one
        assert rel_err(outs[0][k].cpu().numpy(), outs[3][k].cpu().numpy()) < 1e-6
        for key in (0, 3):
            assert rel_err(outs[key][k].cpu().numpy(), ref[k]) < TOL


@pytest.mark.parametrize("M,N", [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10), (63, 120), (31, 1), (200, 333)])
def test_small_gemm_bodies_bitwise_equal(psgd, hip_lib, stage_kernels, M, N):
    """The 32 x 32-tile products have two bodies (psgd_kron_set_tuning(3, .)): same tiles, same K order, same fp32 MFMA
    chains -> bitwise equal updates and applies (single and batched calls)."""
    rng = np.random.default_rng(11 * M + N)
    Ql, Qr = _dev(_tri_factor(rng, M) * 1.5), _dev(_tri_factor(rng, N))
    dX = _dev(rng.standard_normal((M, N)))
    dG = _dev(rng.standard_normal((M, N)) * 2.0)
    outs = []
    try:
        for k in (1, 0):
            hip_lib.psgd_kron_set_tuning(3, k)
            a = psgd.precond_grad_kron(Ql, Qr, dG)
            u = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
            ab = psgd.precond_grad_kron_batched([Ql], [Qr], [dG])[0]
            ub = psgd.update_precond_kron_batched([Ql], [Qr], [dX], [dG], 0.01)[0]
            outs.append((a, u[0], u[1], ab, ub[0], ub[1]))
    finally:
        hip_lib.psgd_kron_set_tuning(3, 1)
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    ref = orc.precond_grad_kron(*(t.cpu().numpy().astype(np.float64) for t in (Ql, Qr, dG)))
    assert rel_err(outs[0][0].cpu().numpy(), ref) < TOL and rel_err(outs[0][3].cpu().numpy(), ref) < TOL


def test_small_update_propagates_nan(psgd):
    """NaN in the data reaches both new factors (tf.reduce_max semantics of psgd.py:177-178)."""
    rng = np.random.default_rng(3)
    M, N = 121, 84
    Ql, Qr = _tri_factor(rng, M), _tri_factor(rng, N)
    dX, dG = rng.standard_normal((M, N)), rng.standard_normal((M, N))
    dXn = dX.copy()
    dXn[5, 7] = np.nan
    bad = psgd.update_precond_kron(_dev(Ql), _dev(Qr), _dev(dXn), _dev(dG), 0.01)
    assert torch.isnan(bad[0]).any() and torch.isnan(bad[1]).any()


# ----------------------------------------------------------------------------- bf16-operand update
BF16_UPD_TOL = 2e-2        # the stated bf16 bar, on the update INCREMENT Q_new - Q_balanced (the quantity the bf16 GEMMs produce)
BF16_UPD_STATE_TOL = 2e-4  # on the factors themselves: increment error x step (0.01) + fp32 rounding


@pytest.mark.parametrize("M,N", [(128, 128), (256, 512), (512, 256), (136, 264), (8, 8), (200, 72), (72, 200), (64, 1000),
                                 (1024, 768), (1160, 520), (2304, 2048),             # (2304 x 2048: the solves through inverses)
                                 (260, 133), (5, 12), (1027, 515), (85, 10)])        # not multiples of 8: padded
def test_dense_dense_update_bf16(psgd, M, N):
    rng = np.random.default_rng(5 * M + N)
    Ql, Qr = (_tri_factor(rng, M) * 3.0).astype(np.float32), _tri_factor(rng, N).astype(np.float32)
    dX = rng.standard_normal((M, N))
    Hl = np.eye(M) + 0.1 * np.diag(rng.uniform(0, 5, M))
    Hr = np.eye(N) + 0.1 * np.diag(rng.uniform(0, 5, N))
    dXb = torch.from_numpy(dX.astype(np.float32)).cuda().to(torch.bfloat16)
    dGb = torch.from_numpy((Hl @ dX @ Hr).astype(np.float32)).cuda().to(torch.bfloat16)
    Ql_new, Qr_new = psgd.update_precond_kron(_dev(Ql), _dev(Qr), dXb, dGb, 0.01)
    assert Ql_new.dtype == torch.float32 and Qr_new.dtype == torch.float32
    f64 = lambda t: t.float().cpu().numpy().astype(np.float64)
    rl, rr = orc.update_precond_kron(Ql.astype(np.float64), Qr.astype(np.float64), f64(dXb), f64(dGb), 0.01)
    assert rel_err(Ql_new.cpu().numpy(), rl) < BF16_UPD_STATE_TOL
    assert rel_err(Qr_new.cpu().numpy(), rr) < BF16_UPD_STATE_TOL
    # the increments (reference: psgd.py:179-180 subtracts from the balanced factors, :166-170)
    rho = np.sqrt(np.max(np.diag(Ql.astype(np.float64))) / np.max(np.diag(Qr.astype(np.float64))))
    assert rel_err(Ql_new.cpu().numpy().astype(np.float64) - Ql / rho, rl - Ql / rho) < BF16_UPD_TOL
    assert rel_err(Qr_new.cpu().numpy().astype(np.float64) - Qr * rho, rr - Qr * rho) < BF16_UPD_TOL
    # factors stay upper triangular, and the fp32 HIP update on the same data agrees to the same bars
    assert torch.equal(Ql_new, torch.triu(Ql_new)) and torch.equal(Qr_new, torch.triu(Qr_new))
    Ql32, Qr32 = psgd.update_precond_kron(_dev(Ql), _dev(Qr), dXb.float(), dGb.float(), 0.01)
    assert rel_err(Ql_new.cpu().numpy(), Ql32.cpu().numpy()) < BF16_UPD_STATE_TOL
    assert rel_err(Qr_new.cpu().numpy(), Qr32.cpu().numpy()) < BF16_UPD_STATE_TOL


@pytest.mark.parametrize("M,N", [(256, 512), (768, 768), (1024, 256), (1280, 1024), (2304, 2048), (2304, 2304)])
def test_bf16_update_stream_k_products(psgd, hip_lib, M, N):
    """The bf16-operand update's gradient products as stream-K launches (k_hgemm_sk_256 + k_hgemm_sk_fix; by default from ~2048 on,
    here forced for every shape the kernel takes -- mode 2: whole-tile rounds + ranges of K tiles for the rest when M = N, ranges only
    otherwise; mode 3: ranges only: a few K tiles each, tiles cut several times).  Against the fp64 oracle on the bf16-rounded data;
    against the one-tile-per-workgroup kernels (same bf16 products, another fp32 summation order); reproducible bit for bit from call
    to call, also with a second stream keeping the CUs busy (no workgroup waits for another)."""
    rng = np.random.default_rng(3 * M + N)
    Ql, Qr = (_tri_factor(rng, M) * 2.0).astype(np.float32), _tri_factor(rng, N).astype(np.float32)
    dX = rng.standard_normal((M, N))
    dXb = torch.from_numpy(dX.astype(np.float32)).cuda().to(torch.bfloat16)
    dGb = torch.from_numpy((dX * np.exp(rng.uniform(-1, 1, (M, 1))) * np.exp(rng.uniform(-1, 1, (1, N)))).astype(np.float32)).cuda().to(torch.bfloat16)
    f64 = lambda t: t.float().cpu().numpy().astype(np.float64)
    res = {}
    try:
        for mode in (0, 3, 2):
            hip_lib.psgd_kron_bf16_set_tuning(4, mode)
            res[mode] = psgd.update_precond_kron(_dev(Ql), _dev(Qr), dXb, dGb, 0.01)
            torch.cuda.synchronize()
        side = torch.cuda.Stream()
        A = torch.randn(4096, 4096, device="cuda")
        with torch.cuda.stream(side):
            for _ in range(10):
                A = torch.mm(A, A) * 1e-4
        again = [psgd.update_precond_kron(_dev(Ql), _dev(Qr), dXb, dGb, 0.01) for _ in range(3)]
        torch.cuda.synchronize()
    finally:
        hip_lib.psgd_kron_bf16_set_tuning(4, 1)
    rl, rr = orc.update_precond_kron(Ql.astype(np.float64), Qr.astype(np.float64), f64(dXb), f64(dGb), 0.01)
    rho = np.sqrt(np.max(np.abs(Ql.astype(np.float64))) / np.max(np.abs(Qr.astype(np.float64))))
    for i, (ref, q0) in enumerate(((rl, Ql.astype(np.float64) / rho), (rr, Qr.astype(np.float64) * rho))):
        for mode in (0, 2, 3):
            got = f64(res[mode][i])
            assert rel_err(got, ref) < BF16_UPD_STATE_TOL
            assert rel_err(got - q0, ref - q0) < BF16_UPD_TOL
        for mode in (2, 3):
            assert rel_err(f64(res[mode][i]) - q0, f64(res[0][i]) - q0) < 1e-3   # same bf16 products, fp32 summation order only
        assert torch.equal(res[2][i], torch.triu(res[2][i]))
        for f in again:
            assert torch.equal(f[i], res[2][i])


@pytest.mark.parametrize("M,N", [(260, 133), (85, 10), (1027, 515)])
@pytest.mark.parametrize("scale", [100.0, 0.01])
def test_bf16_padded_update_with_unbalanced_factors(psgd, M, N, scale):
    """Shapes that are not multiples of 8 run zero-padded.  The pad diagonal goes through the balance of psgd.py:166-170
    and is inverted by the solves of :174: with tiny on it, max diag(Ql) / max diag(Qr) outside (1/16, 16) overflowed
    the inverse and every entry of both new factors came back NaN.  Ql scaled by 100 and by 0.01."""
    rng = np.random.default_rng(7 * M + N)
    Ql, Qr = (_tri_factor(rng, M) * scale).astype(np.float32), _tri_factor(rng, N).astype(np.float32)
    dX = rng.standard_normal((M, N))
    dXb = torch.from_numpy(dX.astype(np.float32)).cuda().to(torch.bfloat16)
    dGb = torch.from_numpy((dX * np.exp(rng.uniform(-1, 1, (M, 1)))).astype(np.float32)).cuda().to(torch.bfloat16)
    Ql_new, Qr_new = psgd.update_precond_kron(_dev(Ql), _dev(Qr), dXb, dGb, 0.01)
    assert Ql_new.shape == (M, M) and Qr_new.shape == (N, N)
    assert torch.isfinite(Ql_new).all() and torch.isfinite(Qr_new).all()
    f64 = lambda t: t.float().cpu().numpy().astype(np.float64)
    rl, rr = orc.update_precond_kron(Ql.astype(np.float64), Qr.astype(np.float64), f64(dXb), f64(dGb), 0.01)
    assert rel_err(Ql_new.cpu().numpy(), rl) < BF16_UPD_STATE_TOL
    assert rel_err(Qr_new.cpu().numpy(), rr) < BF16_UPD_STATE_TOL
    rho = np.sqrt(np.max(np.abs(Ql.astype(np.float64))) / np.max(np.abs(Qr.astype(np.float64))))
    assert rel_err(f64(Ql_new) - Ql / rho, rl - Ql / rho) < BF16_UPD_TOL
    assert rel_err(f64(Qr_new) - Qr * rho, rr - Qr * rho) < BF16_UPD_TOL


@pytest.mark.parametrize("M,N", [(1104, 528), (2048, 1536)])
def test_bf16_update_solves_with_ill_conditioned_factors(psgd, hip_lib, M, N):
    """The bf16-operand update keeps its triangular solves (psgd.py:174) in fp32 except for the trailing PRODUCTS between
    strips, which by default keep three of the six bf16 x 3 split terms (2^-16 per product; tuning key 3).  With
    cond(Q) ~ 1e4 factors (the conditioning multiplies whatever the products lose) the default must agree with the
    all-six-terms solve far inside the bf16 bars, and both with the fp64 oracle on the bf16-rounded data."""
    rng = np.random.default_rng(M + 13 * N)

    def illcond(n):
        d = np.exp(np.linspace(0.0, -np.log(1e4), n))
        rng.shuffle(d)
        return np.triu(rng.standard_normal((n, n)) * (0.3 / n ** 0.5), 1) * d[None, :] + np.diag(d)
    Ql, Qr = illcond(M).astype(np.float32), illcond(N).astype(np.float32)
    assert 3e3 < np.linalg.cond(Ql.astype(np.float64)) < 1e6 and 3e3 < np.linalg.cond(Qr.astype(np.float64)) < 1e6
    dX = rng.standard_normal((M, N))
    dG = np.linalg.solve(Ql.T.astype(np.float64) @ Ql, dX) @ np.linalg.inv(Qr.T.astype(np.float64) @ Qr) \
        * np.exp(rng.uniform(-0.5, 0.5, (1, N)))
    dXb = torch.from_numpy(dX.astype(np.float32)).cuda().to(torch.bfloat16)
    dGb = torch.from_numpy(dG.astype(np.float32)).cuda().to(torch.bfloat16)
    f64 = lambda t: t.float().cpu().numpy().astype(np.float64)
    res = {}
    try:
        for lite in (1, 0):
            hip_lib.psgd_kron_bf16_set_tuning(3, lite)
            res[lite] = [f64(t) for t in psgd.update_precond_kron(_dev(Ql), _dev(Qr), dXb, dGb, 0.01)]
    finally:
        hip_lib.psgd_kron_bf16_set_tuning(3, 1)
    rl, rr = orc.update_precond_kron(Ql.astype(np.float64), Qr.astype(np.float64), f64(dXb), f64(dGb), 0.01)
    rho = np.sqrt(np.max(np.abs(Ql.astype(np.float64))) / np.max(np.abs(Qr.astype(np.float64))))
    for i, (ref, q0) in enumerate(((rl, Ql.astype(np.float64) / rho), (rr, Qr.astype(np.float64) * rho))):
        for lite in (1, 0):
            assert np.isfinite(res[lite][i]).all()
            assert rel_err(res[lite][i], ref) < BF16_UPD_STATE_TOL
            assert rel_err(res[lite][i] - q0, ref - q0) < BF16_UPD_TOL
        # three-term against six-term trailing products: an order of magnitude inside the bf16 increment bar
        assert rel_err(res[1][i] - q0, res[0][i] - q0) < 0.1 * BF16_UPD_TOL


def test_prepared_factor_state_rules(psgd):
    """The reuse of prepared factor-only state (kron.py): version-tracked in-place writes are seen; writes through
    `.data` are not (documented) until invalidate_factor_cache(); set_factor_cache(False) switches the reuse off;
    inference-mode tensors (no version counter) work and never hit."""
    from psgd_tf_amd import kron
    rng = np.random.default_rng(11)
    for M, N, dt in ((257, 120, torch.float32), (1280, 1024, torch.float32), (512, 256, torch.bfloat16)):
        tol = TOL if dt == torch.float32 else 2e-2
        Ql, Qr = _dev(_tri_factor(rng, M)), _dev(_tri_factor(rng, N))
        G = _dev(rng.standard_normal((M, N))).to(dt)
        ref = lambda: orc.precond_grad_kron(*(t.float().cpu().numpy().astype(np.float64) for t in (Ql, Qr, G)))
        o1 = psgd.precond_grad_kron(Ql, Qr, G)
        assert rel_err(o1.float().cpu().numpy(), ref()) < tol
        Ql.mul_(1.5)                                           # version-tracked: seen
        assert rel_err(psgd.precond_grad_kron(Ql, Qr, G).float().cpu().numpy(), ref()) < tol
        Qr.data.mul_(0.5)                                      # bypasses the version counter: stale until invalidated
        kron.invalidate_factor_cache()
        assert rel_err(psgd.precond_grad_kron(Ql, Qr, G).float().cpu().numpy(), ref()) < tol
        old = kron.set_factor_cache(False)
        try:
            assert old is True
            Qr.data.mul_(2.0)                                  # reuse off: every call rebuilds, nothing can be stale
            assert rel_err(psgd.precond_grad_kron(Ql, Qr, G).float().cpu().numpy(), ref()) < tol
            Ql.data.mul_(0.7)
            assert rel_err(psgd.precond_grad_kron(Ql, Qr, G).float().cpu().numpy(), ref()) < tol
        finally:
            kron.set_factor_cache(True)
        with torch.inference_mode():
            Qli, Qri = Ql * 1.0, Qr * 1.0                      # inference tensors: ._version raises
            for _ in range(2):
                oi = psgd.precond_grad_kron(Qli, Qri, G)
        assert rel_err(oi.float().cpu().numpy(), ref()) < tol


def test_apply_captured_in_a_graph_follows_in_graph_factor_updates(psgd):
    """While the stream is being captured the factor-only half is always part of the captured work (no reuse decision
    is frozen into the graph): a graph of [factor update in place; apply] replays correctly."""
    rng = np.random.default_rng(5)
    M, N = 257, 120
    Ql, Qr = _dev(_tri_factor(rng, M)), _dev(_tri_factor(rng, N))
    G = _dev(rng.standard_normal((M, N)))
    out = torch.empty_like(G)
    psgd.precond_grad_kron(Ql, Qr, G)                           # warm: workspace made, Grams prepared for (Ql, Qr)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        psgd.precond_grad_kron(Ql, Qr, G)                       # workspace of the capture stream exists before the capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        Ql.mul_(1.1)
        out.copy_(psgd.precond_grad_kron(Ql, Qr, G))
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        ref = orc.precond_grad_kron(*(t.cpu().numpy().astype(np.float64) for t in (Ql, Qr, G)))
        assert rel_err(out.cpu().numpy(), ref) < TOL


def test_bf16_update_rejects_mixed_dtypes(psgd):
    with pytest.raises(TypeError):
        psgd.update_precond_kron(torch.eye(8, device="cuda"), torch.eye(16, device="cuda"),
                                 torch.ones(8, 16, device="cuda", dtype=torch.bfloat16), torch.ones(8, 16, device="cuda"))
    with pytest.raises(TypeError):      # bf16 factors are not accepted (master copies are fp32)
        psgd.update_precond_kron(torch.eye(8, device="cuda", dtype=torch.bfloat16), torch.eye(16, device="cuda"),
                                 torch.ones(8, 16, device="cuda"), torch.ones(8, 16, device="cuda"))


@pytest.mark.parametrize("M,N", [(257, 120), (85, 10), (16, 40), (640, 1024), (1100, 1030), (2049, 1024), (2304, 2048)])
def test_prepared_grams_follow_the_factors(psgd, hip_lib, M, N):
    """fp32 apply: the Grams of the factors are kept in the workspace and recomputed only when the factors change (same
    rules as the bf16 copies below); single and batched calls; small (both Grams), large (reference-order Gram) and
    pre-split-plane plans (M, N >= 1024: Gram and factor planes are the prepared state)."""
    _prepared_grams_case(psgd, M, N)


def _prepared_grams_case(psgd, M, N):
    rng = np.random.default_rng(M + N)
    Ql, Qr = _dev(_tri_factor(rng, M).astype(np.float32)), _dev(_tri_factor(rng, N).astype(np.float32))
    G, G2 = _dev(rng.standard_normal((M, N))), _dev(rng.standard_normal((M, N)))
    ref = lambda ql, qr, g: orc.precond_grad_kron(*(t.cpu().numpy().astype(np.float64) for t in (ql, qr, g)))
    first = psgd.precond_grad_kron(Ql, Qr, G)                      # (large layers: new factors take the Gram-free chain, another association)
    a = psgd.precond_grad_kron(Ql, Qr, G)                          # the same factors again: their Grams are made ...
    b = psgd.precond_grad_kron(Ql, Qr, G)                          # ... and reused
    c = psgd.precond_grad_kron(Ql, Qr, G2)                         # ... for another gradient too
    assert rel_err(first.cpu().numpy(), ref(Ql, Qr, G)) < TOL
    assert torch.equal(a, b) and rel_err(a.cpu().numpy(), ref(Ql, Qr, G)) < TOL and rel_err(c.cpu().numpy(), ref(Ql, Qr, G2)) < TOL
    # an update in between uses the same workspace: the prepared Grams must survive it
    psgd.update_precond_kron(Ql, Qr, G, G2, 0.01)
    assert torch.equal(psgd.precond_grad_kron(Ql, Qr, G), a)
    Ql.mul_(1.5)                                                   # in place: new version -> recomputed
    assert rel_err(psgd.precond_grad_kron(Ql, Qr, G).cpu().numpy(), ref(Ql, Qr, G)) < TOL
    del Qr
    Qr2 = _dev((_tri_factor(rng, N) * 0.5).astype(np.float32))     # a new tensor, usually at the address just freed
    assert rel_err(psgd.precond_grad_kron(Ql, Qr2, G).cpu().numpy(), ref(Ql, Qr2, G)) < TOL
    # batched
    o1 = psgd.precond_grad_kron_batched([Ql, Ql], [Qr2, Qr2], [G, G2])
    o2 = psgd.precond_grad_kron_batched([Ql, Ql], [Qr2, Qr2], [G2, G])
    assert rel_err(o1[0].cpu().numpy(), ref(Ql, Qr2, G)) < TOL and rel_err(o2[0].cpu().numpy(), ref(Ql, Qr2, G2)) < TOL
    assert torch.equal(o1[1], o2[0]) and torch.equal(o1[0], o2[1])
    Qr2.add_(torch.triu(torch.full_like(Qr2, 0.01)))
    o3 = psgd.precond_grad_kron_batched([Ql, Ql], [Qr2, Qr2], [G, G2])
    assert rel_err(o3[0].cpu().numpy(), ref(Ql, Qr2, G)) < TOL and rel_err(o3[1].cpu().numpy(), ref(Ql, Qr2, G2)) < TOL


@pytest.mark.parametrize("M,N", [(1024, 1024), (1030, 1100), (1024, 2049), (1500, 1027), (1000, 1000), (260, 3100), (8200, 70)])
def test_large_apply_on_operand_planes(psgd, M, N):
    """At least 64 output tiles of 128 x 128 (skinny shapes included: the short side is zero-padded to a tile): the apply
    runs on operands split once into planes (k_gemm_p3; two fp16 planes and a power-of-two scale per matrix by default,
    three bf16 planes -- x = h + m + l exactly, the products of the in-GEMM split -- with tuning key 12 = 0).  Every path
    against the fp64 oracle, shapes that are not multiples of the 128-tile (zero-padded planes), and against each other."""
    from psgd_tf_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(M * 3 + N)
    Ql, Qr = _tri_factor(rng, M).astype(np.float32), _tri_factor(rng, N).astype(np.float32)
    G = rng.standard_normal((M, N)).astype(np.float32)
    ref = orc.precond_grad_kron(Ql.astype(np.float64), Qr.astype(np.float64), G.astype(np.float64))
    outs = []
    try:
        for planes, f16 in ((1, 1), (1, 0), (0, 0)):
            lib.psgd_kron_set_tuning(4, planes)
            lib.psgd_kron_set_tuning(12, f16)
            out = psgd.precond_grad_kron(_dev(Ql), _dev(Qr), _dev(G))          # new factor tensors: prepared state rebuilt
            assert rel_err(out.cpu().numpy(), ref) < TOL
            assert torch.equal(out, psgd.precond_grad_kron(_dev(Ql), _dev(Qr), _dev(G)))
            outs.append(out)
    finally:
        lib.psgd_kron_set_tuning(4, 1)
        lib.psgd_kron_set_tuning(12, 1)
    # (same products; a few-tile shape with a long K -- 8200 x 70 -- sums its K range in chunks on the planes paths)
    assert rel_err(outs[1].cpu().numpy(), outs[2].cpu().numpy()) < 3e-6
    assert rel_err(outs[0].cpu().numpy(), outs[2].cpu().numpy()) < 3e-6


def test_gradient_grid_k_split_is_deterministic_and_equivalent(psgd):
    """M = N = 2944: 2 x 276 upper gradient tiles on 512 block slots leave a last round of 40, so the last 80 tiles are
    split along K over 8 blocks each (tuning key 6): same factors as the unsplit grid to fp32 rounding, bit-identical
    from call to call (the partials are summed in chunk order, not in arrival order)."""
    from psgd_tf_amd import _lib
    lib = _lib.load()
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the shape is chosen for 512 block slots")
    M = N = 2944
    g = torch.Generator(device="cuda").manual_seed(11)
    Ql = torch.triu(torch.randn(M, M, device="cuda", generator=g) * 0.02, 1) + torch.eye(M, device="cuda") * 1.5
    Qr = torch.triu(torch.randn(N, N, device="cuda", generator=g) * 0.02, 1) + torch.eye(N, device="cuda")
    dX = torch.randn(M, N, device="cuda", generator=g)
    dG = 1.3 * dX + 0.3 * torch.randn(M, N, device="cuda", generator=g)
    outs = []
    try:
        for split in (1, 1, 0):
            lib.psgd_kron_set_tuning(6, split)
            outs.append(psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01))
    finally:
        lib.psgd_kron_set_tuning(6, 1)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    rho = torch.sqrt(Ql.diagonal().max() / Qr.diagonal().max())
    for a, b, base in ((outs[0][0], outs[2][0], Ql / rho), (outs[0][1], outs[2][1], Qr * rho)):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
        assert rel_err((a - base).cpu().numpy(), (b - base).cpu().numpy()) < 1e-4


@pytest.mark.parametrize("M,N", [(1024, 1024), (1030, 1100)])
def test_large_update_planes_against_in_gemm_split(psgd, M, N):
    """The plane path of the update (tuning key 4) and the in-GEMM split produce the same factors to fp32 rounding (the
    fp64 oracle comparison of both is test_dense_dense_update / PLANE_SHAPES with the default path)."""
    from psgd_tf_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(M + 7 * N)
    Ql, Qr = _dev(_tri_factor(rng, M) * 2.0), _dev(_tri_factor(rng, N))
    dX = _dev(rng.standard_normal((M, N)))
    dG = _dev(rng.standard_normal((M, N)) * 0.5) + 1.3 * dX
    outs = []
    try:
        for planes in (1, 0):
            lib.psgd_kron_set_tuning(4, planes)
            outs.append(psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01))
    finally:
        lib.psgd_kron_set_tuning(4, 1)
    rho = torch.sqrt(Ql.diagonal().max() / Qr.diagonal().max())
    for a, b, base in ((outs[0][0], outs[1][0], Ql / rho), (outs[0][1], outs[1][1], Qr * rho)):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
        assert rel_err((a - base).cpu().numpy(), (b - base).cpu().numpy()) < 1e-4
        assert torch.equal(a, torch.triu(a))


@pytest.mark.parametrize("M,N", [(1030, 1100), (260, 3100), (2200, 1300)])
def test_plane_formats_of_the_large_update_agree(psgd, M, N):
    """Tuning key 12: the large update on f16 x 2 planes (2, default), with only the apply on them (1) and on bf16 x 3
    planes (0) -- same factors to fp32 rounding, each inside the bars of test_dense_dense_update against the fp64
    oracle.  (2200 x 1300: the solves' update products run on the factors' planes too, tuning keys 13-15.)"""
    from psgd_tf_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5 * M + N)
    a32 = [a.astype(np.float32) for a in (_tri_factor(rng, M) * 2.0, _tri_factor(rng, N), rng.standard_normal((M, N)))]
    a32.append((a32[2] * np.exp(rng.uniform(-1, 1, (M, 1))) * np.exp(rng.uniform(-1, 1, (1, N)))).astype(np.float32))
    ref = orc.update_precond_kron(*(a.astype(np.float64) for a in a32), 0.01)
    rho = np.sqrt(np.max(np.diag(a32[0])) / np.max(np.diag(a32[1])))
    base = (a32[0].astype(np.float64) / rho, a32[1].astype(np.float64) * rho)
    dev = [_dev(a) for a in a32]
    outs = []
    try:
        for fmt, exact in ((2, 1), (1, 1), (0, 1), (2, 0)):     # (2, 0): chained products write their planes at bound scales (key 16)
            lib.psgd_kron_set_tuning(12, fmt)
            lib.psgd_kron_set_tuning(16, exact)
            out = psgd.update_precond_kron(*dev, 0.01)
            again = psgd.update_precond_kron(*dev, 0.01)
            assert torch.equal(out[0], again[0]) and torch.equal(out[1], again[1])
            for got, r, b in zip(out, ref, base):
                g = got.cpu().numpy().astype(np.float64)
                assert rel_err(g, r) < TOL and rel_err(g - b, r - b) < INCR_TOL
                assert torch.equal(got, torch.triu(got))
            outs.append(out)
    finally:
        lib.psgd_kron_set_tuning(12, 2)
        lib.psgd_kron_set_tuning(16, 1)
    for other in outs[1:]:
        for a, b, bb in zip(outs[0], other, base):
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
            assert rel_err(a.cpu().numpy() - bb, b.cpu().numpy() - bb) < 1e-4


@pytest.mark.parametrize("scale", [1e-37, 1e-25, 1e-8, 1e8, 1e25])
def test_f16_planes_follow_the_magnitude_of_the_data(psgd, scale):
    """f16 x 2 planes carry one power-of-two scale per matrix taken from its actual maximum: data of any fp32 magnitude
    (1e-37: the scale exponent is clamped so that 2^e stays a normal number) gives the accuracy of O(1) data.  The
    update is invariant to a common scale of dX / 1/dG up to the step normalisation, so it is run on dG * s, dX / s."""
    M, N = 1030, 1100
    rng = np.random.default_rng(11)
    Ql, Qr = _tri_factor(rng, M).astype(np.float32), _tri_factor(rng, N).astype(np.float32)
    G = (rng.standard_normal((M, N)) * scale).astype(np.float32)
    ref = orc.precond_grad_kron(Ql.astype(np.float64), Qr.astype(np.float64), G.astype(np.float64))
    out = psgd.precond_grad_kron(_dev(Ql), _dev(Qr), _dev(G))
    assert rel_err(out.cpu().numpy(), ref) < TOL
    if 1e-30 < scale < 1e30:                               # (beyond: dG dX products leave the fp32 range in the reference too)
        s = np.float32(scale) ** np.float32(0.5)
        dX = (rng.standard_normal((M, N)) / s).astype(np.float32)
        dG = (dX.astype(np.float64) * s * s * np.exp(rng.uniform(-1, 1, (M, 1)))).astype(np.float32)
        r = orc.update_precond_kron(*(a.astype(np.float64) for a in (Ql, Qr, dX, dG)), 0.01)
        got = psgd.update_precond_kron(_dev(Ql), _dev(Qr), _dev(dX), _dev(dG), 0.01)
        for g, rr in zip(got, r):
            assert rel_err(g.cpu().numpy(), rr) < TOL


@pytest.mark.parametrize("M,N", [(2304, 2048), (2100, 3000), (1030, 1100), (4096, 4096), (640, 2304)])
def test_tile_scales_agree_with_matrix_scales(psgd, M, N):
    """Round 5, tuning key 28: chained f16 x 2 plane products write their planes from the epilogue with one power-of-two scale per
    128 x 128 TILE (exponents in a table; the consumer's K loop shifts its accumulators when the scale changes) instead of fp32 out +
    max|C| + a split launch with one scale per matrix.  A power-of-two scale does not change a value's bits unless it underflows, so
    the two forms agree to fp32 rounding; both hold the parity bars.  Update (both stream orders of the inverse route, blocks of 1024
    and 2048), apply with new factors, bf16-operand update."""
    from psgd_tf_amd import kron
    rng = np.random.default_rng(7 * M + N)
    a32 = [a.astype(np.float32) for a in (_tri_factor(rng, M, 0.02) * 2.0, _tri_factor(rng, N, 0.02), rng.standard_normal((M, N)))]
    a32.append((a32[2] * np.exp(rng.uniform(-1, 1, (M, 1))) * np.exp(rng.uniform(-1, 1, (1, N)))).astype(np.float32))
    G = rng.standard_normal((M, N)).astype(np.float32)
    dev = [_dev(a) for a in a32]
    check_oracle = M * N <= 2304 * 3000
    if check_oracle:
        ref = orc.update_precond_kron(*(a.astype(np.float64) for a in a32), 0.01)
        ref_a = orc.precond_grad_kron(a32[0].astype(np.float64), a32[1].astype(np.float64), G.astype(np.float64))
    rho = np.sqrt(np.max(np.diag(a32[0])) / np.max(np.diag(a32[1])))
    base = (a32[0].astype(np.float64) / rho, a32[1].astype(np.float64) * rho)
    res = {}
    try:
        for ts, order, blk in ((0, -1, 2048), (1, -1, 2048), (1, 0, 2048), (1, 1, 1024)):
            kron.set_tuning(28, ts); kron.set_tuning(25, order); kron.set_tuning(24, blk)
            up = psgd.update_precond_kron(*dev, 0.01)
            again = psgd.update_precond_kron(*dev, 0.01)
            assert torch.equal(up[0], again[0]) and torch.equal(up[1], again[1])            # run to run: bitwise
            ap = psgd.precond_grad_kron(dev[0].clone(), dev[1].clone(), _dev(G))            # new factor tensors: the Gram-free chain
            ub = psgd.update_precond_kron(dev[0], dev[1], dev[2].bfloat16(), dev[3].bfloat16(), 0.01)
            res[(ts, order, blk)] = (up, ap, ub)
            if check_oracle:
                for got, r, b in zip(up, ref, base):
                    g = got.cpu().numpy().astype(np.float64)
                    assert rel_err(g, r) < TOL and rel_err(g - b, r - b) < INCR_TOL, (ts, order, blk)
                    assert torch.equal(got, torch.triu(got))
                assert rel_err(ap.cpu().numpy(), ref_a) < TOL
    finally:
        kron.set_tuning(28, 1); kron.set_tuning(25, -1); kron.set_tuning(24, 2048)
    first = res[(0, -1, 2048)]
    for key, (up, ap, ub) in res.items():
        for a, b, bb in zip(first[0], up, base):
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6, key
            assert rel_err(a.cpu().numpy() - bb, b.cpu().numpy() - bb) < 1e-4, key
        assert rel_err(first[1].cpu().numpy(), ap.cpu().numpy()) < 1e-6, key
        for a, b in zip(first[2], ub):
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6, key


@pytest.mark.parametrize("direction", ["rising", "falling", "blocks"])
def test_tile_scales_with_magnitudes_that_change_along_k(psgd, direction):
    """The consumer of tile-scaled planes moves its accumulators from one power-of-two scale to the next at 128-k boundaries (exactly:
    v_ldexp_f32).  Data whose magnitude rises / falls by 2^40 along the contraction index of the chained products (rows of G scaled,
    so that K of the left-factor products runs over them), and data with whole all-zero tile rows and columns (exponent `any`: the
    accumulators stay where they are): the apply and the update stay at the accuracy of uniform data."""
    M, N = 2304, 2048
    rng = np.random.default_rng(21)
    Ql, Qr = _tri_factor(rng, M, 0.02).astype(np.float32), _tri_factor(rng, N, 0.02).astype(np.float32)
    G = rng.standard_normal((M, N))
    if direction == "blocks":
        G[256:512, :] = 0.0
        G[:, 1024:1280] = 0.0
        G[1500:1700, 100:900] *= 1e-6
    else:
        ramp = np.exp(np.linspace(-14, 14, M))
        G = G * (ramp if direction == "rising" else ramp[::-1])[:, None]
    G = G.astype(np.float32)
    ref = orc.precond_grad_kron(Ql.astype(np.float64), Qr.astype(np.float64), G.astype(np.float64))
    out = psgd.precond_grad_kron(_dev(Ql), _dev(Qr), _dev(G)).cpu().numpy().astype(np.float64)
    assert np.isfinite(out).all() and rel_err(out, ref) < TOL
    assert np.max(np.abs(out - ref)) < TOL * np.max(np.abs(ref))
    dX = (rng.standard_normal((M, N)) * (1.0 if direction == "blocks" else 1.0 / np.exp(np.linspace(-3, 3, M))[:, None])).astype(np.float32)
    dG = (G if direction == "blocks" else G * np.exp(-np.abs(np.linspace(-11, 11, M)))[:, None]).astype(np.float32)
    r = orc.update_precond_kron(*(a.astype(np.float64) for a in (Ql, Qr, dX, dG)), 0.01)
    got = psgd.update_precond_kron(_dev(Ql), _dev(Qr), _dev(dX), _dev(dG), 0.01)
    for g, rr in zip(got, r):
        assert torch.isfinite(g).all() and rel_err(g.cpu().numpy(), rr) < TOL


@pytest.mark.parametrize("tiny", [1e-20, 1e-30])
def test_tile_scales_do_not_overflow_behind_a_tiny_tile(psgd, tiny):
    """moving the accumulators UP to the unit of a much finer K tile multiplied them by 2^(c - cur) with no clamp --
    blocks of the data scaled by 1e-20 / 1e-30 (2^-66 / 2^-100) behind full-size ones gave accumulators of 2^30 * 2^100 = inf.
    Now such a tile is dropped (it is below fp32 resolution of the sum, and the matrix-wide scale of round 4 flushed it to zero
    anyway): finite results at the parity tolerance for the apply and both gradient branches of the update, with the tiny blocks
    placed so that the contraction of every chained product meets them AFTER large tiles (falling) and before (rising)."""
    M, N = 2304, 2048
    rng = np.random.default_rng(33)
    Ql, Qr = _tri_factor(rng, M, 0.02).astype(np.float32), _tri_factor(rng, N, 0.02).astype(np.float32)
    for where in ("tail", "head", "middle"):
        G = rng.standard_normal((M, N))
        rows = {"tail": slice(M - 640, M), "head": slice(0, 640), "middle": slice(900, 1412)}[where]
        cols = {"tail": slice(N - 512, N), "head": slice(0, 512), "middle": slice(700, 1212)}[where]
        G[rows, :] *= tiny
        G[:, cols] *= tiny
        G = G.astype(np.float32)
        ref = orc.precond_grad_kron(Ql.astype(np.float64), Qr.astype(np.float64), G.astype(np.float64))
        out = psgd.precond_grad_kron(_dev(Ql), _dev(Qr), _dev(G)).cpu().numpy().astype(np.float64)
        assert np.isfinite(out).all(), where
        assert rel_err(out, ref) < TOL and np.max(np.abs(out - ref)) < TOL * np.max(np.abs(ref)), where
        dX = rng.standard_normal((M, N)).astype(np.float32)
        for dXv, dGv in ((dX, G), (G, dX)):                  # the tiny blocks in dG (the A chain) and in dX (the solve chain)
            r = orc.update_precond_kron(*(a.astype(np.float64) for a in (Ql, Qr, dXv, dGv)), 0.01)
            got = psgd.update_precond_kron(_dev(Ql), _dev(Qr), _dev(dXv), _dev(dGv), 0.01)
            for g, rr in zip(got, r):
                assert torch.isfinite(g).all() and rel_err(g.cpu().numpy(), rr) < TOL, where


@pytest.mark.parametrize("M,N", [(2048, 2048), (2100, 2304), (2560, 2048)])
def test_fused_prologue_on_a_poisoned_workspace(psgd, hip_lib, M, N):
    """Round 6: on the inverse route the prologue is rho + ONE sweep (k_kron_balance_planes: balanced upper tiles in fp32, both plane forms
    at tile scales) that neither reads nor writes the 128-tiles below the diagonals.  With every byte of the workspace set to 0xFF
    beforehand (NaN in fp32 and in f16) the update still agrees with the fp64 oracle, its results are finite and exactly upper
    triangular, and the round-5 prologue (tuning key 31 = 0: two sweeps, one scale per factor) gives the same factors to rounding."""
    from psgd_tf_amd import kron
    rng = np.random.default_rng(M * 5 + N)
    Ql, Qr = (_tri_factor(rng, M, 0.02) * 2.5).astype(np.float32), _tri_factor(rng, N, 0.02).astype(np.float32)
    dX, dG = rng.standard_normal((M, N)).astype(np.float32), rng.standard_normal((M, N)).astype(np.float32)
    ref = orc.update_precond_kron(*(a.astype(np.float64) for a in (Ql, Qr, dX, dG)), 0.01)
    dev = [_dev(a) for a in (Ql, Qr, dX, dG)]
    outs = {}
    try:
        for key in (1, 0):
            assert hip_lib.psgd_kron_set_tuning(31, key) == 0
            kron._kron_workspace(dev[0].device, M, N).fill_(0xFF)
            got = psgd.update_precond_kron(*dev, 0.01)
            torch.cuda.synchronize()
            for g, r in zip(got, ref):
                q = g.cpu().numpy()
                assert np.isfinite(q).all() and np.array_equal(q, np.triu(q)), key
                assert rel_err(q, r) < TOL, key
            outs[key] = [g.cpu().numpy() for g in got]
    finally:
        hip_lib.psgd_kron_set_tuning(31, 1)
    for a, b in zip(outs[1], outs[0]):
        assert rel_err(a, b.astype(np.float64)) < 2e-6


@pytest.mark.parametrize("M,N", [(2304, 2048), (2048, 2304), (3072, 2560), (4096, 4096), (1536, 2048), (4096, 2048)])
def test_first_call_on_a_poisoned_workspace_equals_the_second(psgd, M, N):
    """Every stream order of the large update (three streams from 2048 on): a consumer that runs ahead of its producer, or reads what
    no launch of the call wrote, shows on the FIRST call on a workspace full of 0xFF bytes and is masked on every later one (same inputs,
    the previous call's intermediates are in place).  First call == second call, bit for bit, and finite."""
    from psgd_tf_amd import kron
    rng = np.random.default_rng(M + 3 * N)
    Ql, Qr = _dev(_tri_factor(rng, M, 0.02) * 1.5), _dev(_tri_factor(rng, N, 0.02))
    dX, dG = _dev(rng.standard_normal((M, N))), _dev(rng.standard_normal((M, N)) * 2.0)
    kron._kron_workspace(Ql.device, M, N).fill_(0xFF)
    first = [t.clone() for t in psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)]
    second = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
    torch.cuda.synchronize()
    for a, b in zip(first, second):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_large_update_propagates_nan_through_tile_scales(psgd):
    """A NaN in the data reaches both new factors on the large path too (a tile that holds one gets scale 1; the values carry it)."""
    M, N = 2048, 2176
    rng = np.random.default_rng(2)
    Ql, Qr = _tri_factor(rng, M, 0.02).astype(np.float32), _tri_factor(rng, N, 0.02).astype(np.float32)
    dX, dG = rng.standard_normal((M, N)).astype(np.float32), rng.standard_normal((M, N)).astype(np.float32)
    dG[700, 900] = np.nan
    got = psgd.update_precond_kron(_dev(Ql), _dev(Qr), _dev(dX), _dev(dG), 0.01)
    assert torch.isnan(got[0]).any() and torch.isnan(got[1]).any()
    G = dX.copy()
    G[3, 5] = np.inf
    out = psgd.precond_grad_kron(_dev(Ql), _dev(Qr), _dev(G))
    assert not torch.isfinite(out).all()


def test_f16_planes_with_a_wide_range_inside_one_matrix(psgd):
    """Elements far below a matrix' maximum: rows of G over e^+-14 and factors whose diagonals span 10^+-3 -- the residual
    plane is stored pre-scaled (M = 2^11 m), so small elements keep their second 11 bits; norm-wise and element-wise
    (relative to the largest element) the result stays at the accuracy of uniform data."""
    M, N = 1100, 1030
    rng = np.random.default_rng(12)
    wide = lambda n: (np.triu(rng.standard_normal((n, n)) * 0.02, 1) * np.exp(np.linspace(-3.45, 3.45, n))[:, None]
                      + np.diag(rng.permutation(np.exp(np.linspace(-3.45, 3.45, n))))).astype(np.float32)
    Ql, Qr = wide(M), wide(N)
    G = (rng.standard_normal((M, N)) * np.exp(np.linspace(-14, 14, M))[:, None]).astype(np.float32)
    ref = orc.precond_grad_kron(Ql.astype(np.float64), Qr.astype(np.float64), G.astype(np.float64))
    out = psgd.precond_grad_kron(_dev(Ql), _dev(Qr), _dev(G)).cpu().numpy().astype(np.float64)
    assert rel_err(out, ref) < TOL
    assert np.max(np.abs(out - ref)) < TOL * np.max(np.abs(ref))
    single = np.zeros((M, N), np.float32)
    single[5, 7] = 3.0
    ref = orc.precond_grad_kron(Ql.astype(np.float64), Qr.astype(np.float64), single.astype(np.float64))
    assert rel_err(psgd.precond_grad_kron(_dev(Ql), _dev(Qr), _dev(single)).cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("M,N", [(2304, 2048), (2100, 3000)])
def test_solves_through_inverses_agree_with_substitution(psgd, M, N):
    """Both factors >= 2048: the solves of psgd.py:174 run as products with explicit inverses (tuning key 11, default) built by
    recursive doubling on the f16 x 2 planes; key 11 = 0 keeps the substitution strips.  Same factors to fp32 rounding, each
    route inside the bars of test_dense_dense_update against the fp64 oracle (2100 x 3000: sizes that are not multiples of the
    128-tile or of a doubling level)."""
    from psgd_tf_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3 * M + N)
    a32 = [a.astype(np.float32) for a in (_tri_factor(rng, M, 0.02) * 2.0, _tri_factor(rng, N, 0.02), rng.standard_normal((M, N)))]
    a32.append((a32[2] * np.exp(rng.uniform(-1, 1, (M, 1))) * np.exp(rng.uniform(-1, 1, (1, N)))).astype(np.float32))
    ref = orc.update_precond_kron(*(a.astype(np.float64) for a in a32), 0.01)
    rho = np.sqrt(np.max(np.diag(a32[0])) / np.max(np.diag(a32[1])))
    base = (a32[0].astype(np.float64) / rho, a32[1].astype(np.float64) * rho)
    dev = [_dev(a) for a in a32]
    outs = []
    try:
        for inv in (1, 0):
            lib.psgd_kron_set_tuning(11, inv)
            out = psgd.update_precond_kron(*dev, 0.01)
            again = psgd.update_precond_kron(*dev, 0.01)
            assert torch.equal(out[0], again[0]) and torch.equal(out[1], again[1])
            for got, r, b in zip(out, ref, base):
                g = got.cpu().numpy().astype(np.float64)
                assert rel_err(g, r) < TOL and rel_err(g - b, r - b) < INCR_TOL
                assert torch.equal(got, torch.triu(got))
            outs.append(out)
    finally:
        lib.psgd_kron_set_tuning(11, 1)
    for a, b, bb in zip(outs[0], outs[1], base):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
        assert rel_err(a.cpu().numpy() - bb, b.cpu().numpy() - bb) < 1e-3


def test_bf16_padded_apply_keeps_its_padded_factors(psgd):
    """A large bf16 apply whose shape is not a multiple of 256 runs zero-padded to one; the padded factors (and with them
    their bf16 copies in the workspace) are kept while the caller's factors are unchanged, and rebuilt when they change."""
    from psgd_tf_amd import kron
    M, N = 1900, 2300
    rng = np.random.default_rng(9)
    Ql, Qr = _dev(_tri_factor(rng, M).astype(np.float32)), _dev(_tri_factor(rng, N).astype(np.float32))
    G = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda().to(torch.bfloat16)
    ref = lambda ql, qr: orc.precond_grad_kron(ql.cpu().numpy().astype(np.float64), qr.cpu().numpy().astype(np.float64),
                                               G.float().cpu().numpy().astype(np.float64))
    a = psgd.precond_grad_kron(Ql, Qr, G)
    key = [k for k in kron._padded_factors if k[1:5] == (M, N, 2048, 2304)]
    assert len(key) == 1
    padded = kron._padded_factors[key[0]][1]
    b = psgd.precond_grad_kron(Ql, Qr, G)
    assert kron._padded_factors[key[0]][1] is padded and torch.equal(a, b)          # reused
    assert rel_err(a.float().cpu().numpy(), ref(Ql, Qr)) < BF16_TOL
    Qr.mul_(0.5)                                                                     # in place: new version -> rebuilt
    c = psgd.precond_grad_kron(Ql, Qr, G)
    assert kron._padded_factors[key[0]][1] is not padded and rel_err(c.float().cpu().numpy(), ref(Ql, Qr)) < BF16_TOL


def test_bf16_factor_copies_follow_the_factors(psgd):
    """The bf16 copies of the factors are cached in the workspace and rebuilt only when the factors change: same tensor
    objects -> reused (same result); modified in place (version counter) -> rebuilt; a different tensor that the
    allocator put at the SAME address -> rebuilt (identity of the object, not of the pointer)."""
    M, N = 512, 256
    rng = np.random.default_rng(3)
    Ql, Qr = _dev(_tri_factor(rng, M).astype(np.float32)), _dev(_tri_factor(rng, N).astype(np.float32))
    G = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda().to(torch.bfloat16)
    ref = lambda ql, qr: orc.precond_grad_kron(ql.cpu().numpy().astype(np.float64), qr.cpu().numpy().astype(np.float64),
                                               G.float().cpu().numpy().astype(np.float64))
    a = psgd.precond_grad_kron(Ql, Qr, G)
    b = psgd.precond_grad_kron(Ql, Qr, G)                          # copies reused
    assert torch.equal(a, b) and rel_err(a.float().cpu().numpy(), ref(Ql, Qr)) < BF16_TOL
    Ql.mul_(1.5)                                                   # in place: same object, same pointer, new version
    c = psgd.precond_grad_kron(Ql, Qr, G)
    assert rel_err(c.float().cpu().numpy(), ref(Ql, Qr)) < BF16_TOL and rel_err(c.float().cpu().numpy(), 2.25 * a.float().cpu().numpy()) < BF16_TOL
    ptr = Qr.data_ptr()
    del Qr
    Qr2 = _dev((_tri_factor(rng, N) * 0.5).astype(np.float32))     # usually lands on the block just freed
    d = psgd.precond_grad_kron(Ql, Qr2, G)
    assert rel_err(d.float().cpu().numpy(), ref(Ql, Qr2)) < BF16_TOL, "stale factor copies (same address: %s)" % (Qr2.data_ptr() == ptr)


@pytest.mark.parametrize("tile_choice,x3", [(1, 1), (2, 1), (2, 0), (3, 1)])
@pytest.mark.parametrize("M,N", [(257, 120), (300, 500), (1100, 530), (128, 128), (640, 256)])
def test_every_gemm_tile_size_forced(psgd, hip_lib, stage_kernels, M, N, tile_choice, x3):
    """The fp32 GEMM picks its tile size (32 / 64 / 128) from the problem size; force each one on small
    and ragged shapes (edge tiles, unaligned leading dimensions) and compare with the oracle.  The 128 tile has two
    bodies: the bf16 x 3 split GEMM (x3 = 1, default) and the exact fp32-MFMA one (x3 = 0)."""
    rng = np.random.default_rng(M + N)
    Ql, Qr = _tri_factor(rng, M) * 2.0, _tri_factor(rng, N)
    dX, dG, G = (rng.standard_normal((M, N)) for _ in range(3))
    a32 = [a.astype(np.float32) for a in (Ql, Qr, dX, dG, G)]
    a64 = [a.astype(np.float64) for a in a32]
    hip_lib.psgd_kron_set_tuning(0, tile_choice)
    hip_lib.psgd_kron_set_tuning(1, x3)
    try:
        out = psgd.precond_grad_kron(_dev(a32[0]), _dev(a32[1]), _dev(a32[4]))
        Ql_n, Qr_n = psgd.update_precond_kron(_dev(a32[0]), _dev(a32[1]), _dev(a32[2]), _dev(a32[3]), 0.01)
        torch.cuda.synchronize()
    finally:
        hip_lib.psgd_kron_set_tuning(0, 0)
        hip_lib.psgd_kron_set_tuning(1, 1)
    assert rel_err(out.cpu().numpy(), orc.precond_grad_kron(a64[0], a64[1], a64[4])) < TOL
    Ql_r, Qr_r = orc.update_precond_kron(a64[0], a64[1], a64[2], a64[3], 0.01)
    assert rel_err(Ql_n.cpu().numpy(), Ql_r) < TOL and rel_err(Qr_n.cpu().numpy(), Qr_r) < TOL


def test_batched_lenet_set_equals_per_layer(psgd, hip_lib):
    """The batched calls must agree with the per-layer results: bitwise for the GEMM-only apply; the update's triangular solves
    may take a different blocking per path, so those agree to rounding."""
    _batched_vs_per_layer(psgd, 0)


def _batched_vs_per_layer(psgd, fused):
    shapes = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10), (63, 120), (31, 1), (1, 1), (3, 3)]   # > 8: two chunks
    rng = np.random.default_rng(5)
    Qls = [_dev(_tri_factor(rng, m) * 2.0) for m, n in shapes]
    Qrs = [_dev(_tri_factor(rng, n)) for m, n in shapes]
    dXs = [_dev(rng.standard_normal((m, n))) for m, n in shapes]
    dGs = [_dev(rng.standard_normal((m, n))) for m, n in shapes]
    Gs = [_dev(rng.standard_normal((m, n))) for m, n in shapes]
    outs = psgd.precond_grad_kron_batched(Qls, Qrs, Gs)
    news = psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01)
    for i, (m, n) in enumerate(shapes):
        one = psgd.precond_grad_kron(Qls[i], Qrs[i], Gs[i])
        if fused:
            assert rel_err(outs[i].cpu().numpy(), one.cpu().numpy()) < 2e-6, (m, n)
        else:
            assert torch.equal(outs[i], one), (m, n)
        a, b = psgd.update_precond_kron(Qls[i], Qrs[i], dXs[i], dGs[i], 0.01)
        assert rel_err(news[i][0].cpu().numpy(), a.cpu().numpy()) < 2e-6, (m, n)
        assert rel_err(news[i][1].cpu().numpy(), b.cpu().numpy()) < 2e-6, (m, n)


# larger sparse-format cases: embedding-like shapes (README.md:54 recommends these formats for large
# embeddings), the blocked triangular solve (dense side > 512) and ragged edges
BIG_FORMATS = [
    ("norm_dense", (2, 1500), (64, 64)), ("dense_norm", (64, 64), (2, 1500)),
    ("dense_scale", (600, 600), (1, 37)), ("scale_dense", (1, 37), (600, 600)),
    ("norm_scale", (2, 1001), (1, 129)), ("scale_norm", (1, 129), (2, 1001)),
    ("norm_dense", (2, 33), (530, 530)), ("dense_scale", (257, 257), (1, 120)),
    # embedding shapes: the dense factor's gradient is a few tiles with K >= 4096 (split-K on operand planes), the column
    # reductions run over thousands of rows (row-blocked), the mirrored formats come in as transposed views
    ("norm_dense", (2, 5000), (300, 300)), ("dense_norm", (300, 300), (2, 5000)),
    ("dense_scale", (260, 260), (1, 4500)), ("scale_dense", (1, 4500), (260, 260)), ("norm_scale", (2, 6000), (1, 70)),
    # a dense factor from 512 on with a data matrix of at least 1M elements: the data-sized products run on f16 x 2 operand
    # planes (sparse_gemm), on the caller's stream and on the side stream
    ("dense_scale", (1024, 1024), (1, 4100)), ("norm_dense", (2, 4000), (1100, 1100)), ("dense_norm", (1536, 1536), (2, 1500)),
    ("scale_dense", (1, 1300), (2048, 2048)),
    # ... and at least 8 vectors per column of the dense factor: its solve runs as a product with the explicit inverse (sparse_solve)
    ("norm_dense", (2, 5200), (600, 600)), ("dense_scale", (640, 640), (1, 5400)), ("dense_norm", (520, 520), (2, 4300)),
    # ... and products big enough for the planes whose maxima the gradient's planes then reuse (sparse_grad_splitk on f16 x 2 planes)
    ("norm_dense", (2, 12000), (640, 640)),
    # ... many vectors, but a product BELOW the planes' threshold (13100 x 544^2 < 4e9): the in-place solve must stay on the
    # substitution strips -- an explicit-inverse product on launch_gemm would read Bt while other tiles overwrite it (515 tiles)
    ("norm_dense", (2, 13100), (544, 544)),
]


@pytest.mark.parametrize("fmt,sl,sr", BIG_FORMATS)
def test_sparse_formats_large(psgd, fmt, sl, sr):
    assert orc.kron_format(sl, sr) == fmt
    rng = np.random.default_rng(sl[1] * 3 + sr[1])
    M, N = sl[1], sr[1]
    Ql, Qr = _factor_for(rng, sl, M), _factor_for(rng, sr, N)
    if sl[0] == sl[1]:
        Ql = Ql * 3.0                                # rho != 1
    dX, G = rng.standard_normal((M, N)), rng.standard_normal((M, N))
    dG = dX * np.exp(rng.uniform(-1, 1, (M, 1))) * np.exp(rng.uniform(-1, 1, (1, N)))
    a32 = [a.astype(np.float32) for a in (Ql, Qr, dX, dG, G)]
    a64 = [a.astype(np.float64) for a in a32]
    tQl, tQr = _dev(a32[0]), _dev(a32[1])
    Ql_r, Qr_r = a64[0], a64[1]
    for it in range(3):                               # a short sequence: factors feed back
        tQl, tQr = psgd.update_precond_kron(tQl, tQr, _dev(a32[2]), _dev(a32[3]), 0.02)
        Ql_r, Qr_r = orc.update_precond_kron(Ql_r, Qr_r, a64[2], a64[3], 0.02)
    assert tuple(tQl.shape) == sl and tuple(tQr.shape) == sr
    assert rel_err(tQl.cpu().numpy(), Ql_r) < 3e-5 and rel_err(tQr.cpu().numpy(), Qr_r) < 3e-5
    out = psgd.precond_grad_kron(_dev(a32[0]), _dev(a32[1]), _dev(a32[4]))
    assert out.shape == (M, N)
    assert rel_err(out.cpu().numpy(), orc.precond_grad_kron(a64[0], a64[1], a64[4])) < TOL
    # inputs are never written
    assert np.array_equal(_dev(a32[0]).cpu().numpy(), a32[0])


def test_batched_update_with_mixed_sizes(psgd):
    """update_precond_kron_batched on a list that mixes small layers with layers above 512 (the LSTM / NMT drivers do): the
    small ones still share their launches, the large ones go one by one; results as the per-layer calls, bit for bit."""
    rng = np.random.default_rng(77)
    shapes = [(26, 6), (700, 530), (151, 16), (85, 10), (1024, 300)]
    Qls = [_dev(_tri_factor(rng, m) * 2.0) for m, n in shapes]
    Qrs = [_dev(_tri_factor(rng, n)) for m, n in shapes]
    dXs = [_dev(rng.standard_normal((m, n))) for m, n in shapes]
    dGs = [_dev(rng.standard_normal((m, n))) for m, n in shapes]
    got = psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01)
    assert len(got) == len(shapes)
    for (a, b), Ql, Qr, x, g in zip(got, Qls, Qrs, dXs, dGs):
        wa, wb = psgd.update_precond_kron(Ql, Qr, x, g, 0.01)
        assert torch.equal(a, wa) and torch.equal(b, wb)
        ra, rb = orc.update_precond_kron(*(t.cpu().numpy().astype(np.float64) for t in (Ql, Qr, x, g)), 0.01)
        assert rel_err(a.cpu().numpy(), ra) < TOL and rel_err(b.cpu().numpy(), rb) < TOL


def _spd_cholesky_factor(rng, n, cond_h):
    """Upper-triangular Q with Q'Q = H^-1 for an SPD H with random eigenvectors and cond(H) = cond_h: what a converged
    PSGD factor looks like (psgd.py:175-179 drives Q'Q towards H^-1); cond(Q) = sqrt(cond_h), genuinely ill-conditioned
    (not a diagonal scaling)."""
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.exp(np.linspace(0.0, np.log(cond_h), n))
    return np.linalg.cholesky((V / lam) @ V.T).T


@pytest.mark.parametrize("cond_q", [1e0, 1e1, 1e2, 1e3, 1e4])
def test_update_over_condition_numbers(psgd, cond_q):
    """The large fp32 update against the fp64 oracle for GENUINELY ill-conditioned factors (Cholesky factors of SPD matrices with
    random eigenvectors, cond(Q) = cond_q for both; not the column scalings of test_update_with_ill_conditioned_factors).
    The oracle runs on the fp32-rounded factors, so what is measured is the error of the solves and products.  States at 1e-5
    throughout.  The increments lose accuracy with the conditioning in ANY fp32 arithmetic (eps x cond through the solves of
    psgd.py:174, amplified by the cancellation in A A' - Bt Bt'): the 2e-3 bar holds up to cond 1e3, and the bar beyond is
    10 x eps x cond = 6e-3 at 1e4 (measured on the device: 4.0e-3)."""
    rng = np.random.default_rng(int(np.log10(cond_q)) + 50)
    M, N = 1024, 1152
    Ql = _spd_cholesky_factor(rng, M, cond_q ** 2).astype(np.float32) if cond_q > 1 else np.eye(M, dtype=np.float32)
    Qr = _spd_cholesky_factor(rng, N, cond_q ** 2).astype(np.float32) if cond_q > 1 else np.eye(N, dtype=np.float32)
    Ql /= np.max(np.abs(Ql)); Qr /= np.max(np.abs(Qr))
    dX = rng.standard_normal((M, N)).astype(np.float32)
    Ql64, Qr64 = Ql.astype(np.float64), Qr.astype(np.float64)
    dG = (np.linalg.solve(Ql64.T @ Ql64, dX) @ np.linalg.inv(Qr64.T @ Qr64) * np.exp(rng.uniform(-0.5, 0.5, (1, N)))).astype(np.float32)
    out = [t.cpu().numpy().astype(np.float64) for t in psgd.update_precond_kron(_dev(Ql), _dev(Qr), _dev(dX), _dev(dG), 0.01)]
    ref = orc.update_precond_kron(Ql64, Qr64, dX.astype(np.float64), dG.astype(np.float64), 0.01)
    rho = np.sqrt(np.max(np.abs(Ql64)) / np.max(np.abs(Qr64)))
    bar = max(INCR_TOL, 10 * 6e-8 * cond_q)
    for i, q0 in enumerate((Ql64 / rho, Qr64 * rho)):
        assert np.isfinite(out[i]).all()
        assert rel_err(out[i], ref[i]) < TOL, (i, cond_q)
        assert rel_err(out[i] - q0, ref[i] - q0) < bar, (i, cond_q, bar)


@pytest.mark.parametrize("cond_q", [1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6])
def test_update_over_condition_numbers_inverse_route(psgd, hip_lib, cond_q):
    """The same sweep at a shape whose solves run as products with EXPLICIT INVERSES (M, N >= 2048: tri_inverse), up to the
    conditioning that is ordinary for a converged PSGD factor (cond(Q)^2 = cond(P): 1e5 - 1e6).  States at 1e-5, increments at
    max(2e-3, 10 eps cond), and at every cond the inverse route's errors stay within 2x of the substitution strips' on the very
    same inputs (psgd_kron_set_tuning(11, 0)) -- the route has no condition estimate and no fallback, so this is its licence."""
    rng = np.random.default_rng(int(np.log10(cond_q)) + 70)
    M, N = 2048, 2304
    Ql = _spd_cholesky_factor(rng, M, cond_q ** 2).astype(np.float32) if cond_q > 1 else np.eye(M, dtype=np.float32)
    Qr = _spd_cholesky_factor(rng, N, cond_q ** 2).astype(np.float32) if cond_q > 1 else np.eye(N, dtype=np.float32)
    Ql /= np.max(np.abs(Ql)); Qr /= np.max(np.abs(Qr))
    dX = rng.standard_normal((M, N)).astype(np.float32)
    Ql64, Qr64 = Ql.astype(np.float64), Qr.astype(np.float64)
    dG = (np.linalg.solve(Ql64.T @ Ql64, dX) @ np.linalg.inv(Qr64.T @ Qr64) * np.exp(rng.uniform(-0.5, 0.5, (1, N)))).astype(np.float32)
    ref = orc.update_precond_kron(Ql64, Qr64, dX.astype(np.float64), dG.astype(np.float64), 0.01)
    rho = np.sqrt(np.max(np.abs(Ql64)) / np.max(np.abs(Qr64)))
    bases = (Ql64 / rho, Qr64 * rho)
    args = (_dev(Ql), _dev(Qr), _dev(dX), _dev(dG))

    outs = []

    def errors():
        out = [t.cpu().numpy().astype(np.float64) for t in psgd.update_precond_kron(*args, 0.01)]
        assert all(np.isfinite(o).all() for o in out)
        outs.append(out)
        return [(rel_err(out[i], ref[i]), rel_err(out[i] - bases[i], ref[i] - bases[i])) for i in range(2)]
    e_inv = errors()
    hip_lib.psgd_kron_set_tuning(11, 0)
    try:
        e_sub = errors()
    finally:
        hip_lib.psgd_kron_set_tuning(11, 1)
    if 1e1 <= cond_q <= 1e3:            # (the switch switches: two algorithms for the solves do not agree to the last bit)
        assert any(not np.array_equal(a, b) for a, b in zip(*outs))
    for i in range(2):
        # Near the fixed point the gradient is a difference of two Grams of products through the solves: from cond ~1e5 on fp32
        # loses it whatever the solve (measured at 1e5: the left increment is 1.1e-6 of the factor and 82 % wrong on BOTH routes,
        # the factor itself 9e-7 from fp64), so beyond the stated bar the licence is "no worse than substitution"
        # (at 1e6 the factors themselves are 7e-5 from fp64 on both routes, which return the same bits: the step is too small to
        #  move an fp32 factor)
        bar = max(INCR_TOL, 10 * 6e-8 * cond_q)
        assert e_inv[i][0] < TOL or e_inv[i][0] <= 1.05 * e_sub[i][0], (i, cond_q, e_inv, e_sub)
        assert e_inv[i][1] < bar or e_inv[i][1] <= 1.05 * e_sub[i][1], (i, cond_q, e_inv, e_sub, bar)
        assert e_inv[i][0] <= 2 * e_sub[i][0] + 1e-7 and e_inv[i][1] <= 2 * e_sub[i][1] + 1e-5, (i, cond_q, e_inv, e_sub)


@pytest.mark.parametrize("M,N", [(300, 200), (1100, 530), (2048, 1536), (2176, 2048), (1024, 2560), (3072, 1152), (512, 2816), (4096, 320)])
def test_update_with_ill_conditioned_factors(psgd, M, N):
    """Factors with cond(Q) ~ 1e4 (diagonals spread over four decades, dense upper triangles): the two triangular solves of
    psgd.py:174 carry the conditioning.  Updated factors within 1e-5 of the fp64 oracle, increments within 2e-3 -- the
    bar any faster (e.g. inverse-based) solve has to keep."""
    rng = np.random.default_rng(M + 13 * N)

    def illcond(n):
        d = np.exp(np.linspace(0.0, -np.log(1e4), n))
        rng.shuffle(d)
        Q = np.triu(rng.standard_normal((n, n)) * (0.3 / n ** 0.5), 1) * d[None, :] + np.diag(d)
        return Q
    Ql, Qr = illcond(M), illcond(N)
    assert 3e3 < np.linalg.cond(Ql) < 1e6 and 3e3 < np.linalg.cond(Qr) < 1e6
    dX = rng.standard_normal((M, N))
    dG = np.linalg.solve(Ql.T @ Ql, dX) @ np.linalg.inv(Qr.T @ Qr) * np.exp(rng.uniform(-0.5, 0.5, (1, N)))   # near the fixed point: A ~ Bt
    a32 = [x.astype(np.float32) for x in (Ql, Qr, dX, dG)]
    a64 = [x.astype(np.float64) for x in a32]
    got = psgd.update_precond_kron(*(_dev(x) for x in a32), 0.01)
    ref = orc.update_precond_kron(*a64, 0.01)
    rho = np.sqrt(np.max(np.diag(a64[0])) / np.max(np.diag(a64[1])))
    for g_, r_, q0 in zip(got, ref, (a64[0] / rho, a64[1] * rho)):
        g64 = g_.cpu().numpy().astype(np.float64)
        assert rel_err(g64, r_) < TOL
        assert rel_err(g64 - q0, r_ - q0) < INCR_TOL


def test_lenet5_example_whitens_through_a_replayed_graph(hip_lib):
    """examples/lenet5_kron_step.py: the reference's per-layer call pattern (mnist_with_lenet5.py:51,53) captured once in a CUDA
    graph and replayed on static buffers -- the factors it produces precondition the synthetic Kronecker Hessians."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "lenet5_kron_step.py")
    spec = importlib.util.spec_from_file_location("lenet5_kron_step", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    worst0, worst = mod.main(250)
    assert np.isfinite(worst) and worst < 0.6 * worst0, (worst0, worst)


def test_layer_streams_block_equals_serial_calls(psgd):
    """kron.layer_streams: the per-layer calls of mnist_with_lenet5.py:51,53 on forked streams (eager and as a captured graph with
    one branch per layer) return bit for bit what the same calls return one after the other; later work on the caller's stream sees
    the results (the block joins)."""
    from psgd_tf_amd import kron
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10), (257, 120), (640, 300)]    # (one shape twice: separate workspaces)
    Qs = [(torch.triu(torch.randn(m, m, device="cuda", generator=g)) * 0.1 + torch.eye(m, device="cuda"),
           torch.triu(torch.randn(n, n, device="cuda", generator=g)) * 0.1 + torch.eye(n, device="cuda")) for m, n in shapes]
    dXs = [torch.randn(m, n, device="cuda", generator=g) for m, n in shapes]
    dGs = [torch.randn(m, n, device="cuda", generator=g) for m, n in shapes]
    ref_u = [psgd.update_precond_kron(ql, qr, dx, dg, 0.02) for (ql, qr), dx, dg in zip(Qs, dXs, dGs)]
    ref_a = [psgd.precond_grad_kron(ql, qr, dg) for (ql, qr), dg in zip(Qs, dGs)]
    for n_streams in (8, 3):
        with kron.layer_streams(n_streams):
            got_u = [psgd.update_precond_kron(ql, qr, dx, dg, 0.02) for (ql, qr), dx, dg in zip(Qs, dXs, dGs)]
        sums = [a.sum() + b.sum() for a, b in got_u]                  # (consumed on the caller's stream right after the block)
        with kron.layer_streams(n_streams):
            got_a = [psgd.precond_grad_kron(ql, qr, dg) for (ql, qr), dg in zip(Qs, dGs)]
        for (a, b), (c, d), s_ in zip(got_u, ref_u, sums):
            assert torch.equal(a, c) and torch.equal(b, d)
            assert torch.equal(s_, c.sum() + d.sum())
        for a, c in zip(got_a, ref_a):
            assert torch.equal(a, c)
    with pytest.raises(RuntimeError):
        with kron.layer_streams():
            with kron.layer_streams():
                pass
    # captured: one branch per call
    outs = [torch.empty_like(x) for x in dGs]

    def fn():
        with kron.layer_streams():
            res = [psgd.precond_grad_kron(ql, qr, dg) for (ql, qr), dg in zip(Qs, dGs)]
        for o, r_ in zip(outs, res):
            o.copy_(r_)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for o in outs:
        o.zero_()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    for o, c in zip(outs, ref_a):
        assert torch.equal(o, c)


def test_layer_streams_with_more_calls_than_streams_and_inline_arguments(psgd):
    """with more calls than pool streams, call number streams + 1 onward ran on a stream that had waited for the
    caller's stream only once, at its first use -- an argument computed inside the comprehension (on the caller's stream) could be
    read before it was written, and a temporary dropped at the end of the call could be handed out again while still in use.
    Now every call waits and its arguments are held until the block joins: 12 layers on 2 streams, every argument a temporary."""
    from psgd_tf_amd import kron
    g = torch.Generator(device="cuda").manual_seed(23)
    shapes = [(151, 16), (257, 120), (121, 84), (85, 10), (300, 200), (26, 6)] * 2
    Qs = [(torch.triu(torch.randn(m, m, device="cuda", generator=g)) * 0.1 + torch.eye(m, device="cuda"),
           torch.triu(torch.randn(n, n, device="cuda", generator=g)) * 0.1 + torch.eye(n, device="cuda")) for m, n in shapes]
    Gs = [torch.randn(m, n, device="cuda", generator=g) for m, n in shapes]
    big = torch.randn(4096, 4096, device="cuda", generator=g)
    ref = [psgd.precond_grad_kron(ql, qr, (gr * 1.5 + 0.25)) for (ql, qr), gr in zip(Qs, Gs)]
    ref_u = [psgd.update_precond_kron(ql, qr, gr * 0.5, gr * 1.5 + 0.25, 0.02) for (ql, qr), gr in zip(Qs, Gs)]
    torch.cuda.synchronize()
    for rep in range(5):
        junk = big @ big                                                  # the caller's stream is busy when the block starts
        with kron.layer_streams(2):
            got = [psgd.precond_grad_kron(ql, qr, (gr * 1.5 + 0.25)) for (ql, qr), gr in zip(Qs, Gs)]
        with kron.layer_streams(2):
            got_u = [psgd.update_precond_kron(ql, qr, gr * 0.5, gr * 1.5 + 0.25, 0.02) for (ql, qr), gr in zip(Qs, Gs)]
        for a, c in zip(got, ref):
            assert torch.equal(a, c), rep
        for (a, b), (c, d) in zip(got_u, ref_u):
            assert torch.equal(a, c) and torch.equal(b, d), rep
        del junk


def test_layer_batch_block_equals_the_batched_entry_points(psgd):
    """kron.layer_batch: the reference's per-layer list comprehensions (mnist_with_lenet5.py:51, :53) inside the block run as ONE
    batched call per kind when the block ends -- bit for bit the batched entry points' results, eager and captured; a layer above 512,
    another format and bf16 operands inside the block run at once, in order; update -> apply with the NEW factors in one block works."""
    from psgd_tf_amd import kron
    rng = np.random.default_rng(17)
    shapes = [(26, 6), (151, 16), (257, 120), (121, 84), (85, 10), (63, 120), (31, 1), (1, 1), (3, 3), (500, 512)]
    Qls = [_dev(_tri_factor(rng, m) * 2.0) for m, n in shapes]
    Qrs = [_dev(_tri_factor(rng, n)) for m, n in shapes]
    dXs = [_dev(rng.standard_normal((m, n))) for m, n in shapes]
    dGs = [_dev(rng.standard_normal((m, n))) for m, n in shapes]
    Gs = [_dev(rng.standard_normal((m, n))) for m, n in shapes]
    want_a = psgd.precond_grad_kron_batched(Qls, Qrs, Gs)
    want_u = psgd.update_precond_kron_batched(Qls, Qrs, dXs, dGs, 0.01)
    with kron.layer_batch():
        got_u = [psgd.update_precond_kron(a, b, x, g, 0.01) for a, b, x, g in zip(Qls, Qrs, dXs, dGs)]        # :51
    with kron.layer_batch():
        got_a = [psgd.precond_grad_kron(a, b, g) for a, b, g in zip(Qls, Qrs, Gs)]                            # :53
    for (a, b), (c, d) in zip(got_u, want_u):
        assert torch.equal(a, c) and torch.equal(b, d)
    for a, c in zip(got_a, want_a):
        assert torch.equal(a, c)
    # one block for both comprehensions: the applies see the updated factors (a call of the other kind flushes the queue)
    with kron.layer_batch():
        new = [psgd.update_precond_kron(a, b, x, g, 0.01) for a, b, x, g in zip(Qls, Qrs, dXs, dGs)]
        pre = [psgd.precond_grad_kron(a, b, g) for (a, b), g in zip(new, Gs)]
    want_pre = psgd.precond_grad_kron_batched([a for a, _ in want_u], [b for _, b in want_u], Gs)
    for a, c in zip(pre, want_pre):
        assert torch.equal(a, c)
    # a large layer, a sparse format and bf16 operands inside a block: they run at once, the small ones around them batched
    M = 640
    QlB, QrB = _dev(_tri_factor(rng, M)), _dev(_tri_factor(rng, 300))
    GB = _dev(rng.standard_normal((M, 300)))
    qn = _dev(np.stack([np.ones(77), np.zeros(77)]) + 0.01 * rng.standard_normal((2, 77)))
    Gn = _dev(rng.standard_normal((26, 77)))
    ref_big = psgd.precond_grad_kron(QlB.clone(), QrB.clone(), GB)
    ref_nd = psgd.precond_grad_kron(Qls[0], qn, Gn)
    ref_bf = psgd.precond_grad_kron(Qls[2], Qrs[2], Gs[2].bfloat16())
    with kron.layer_batch():
        mixed = [psgd.precond_grad_kron(Qls[0], Qrs[0], Gs[0]), psgd.precond_grad_kron(QlB.clone(), QrB.clone(), GB),
                 psgd.precond_grad_kron(Qls[1], Qrs[1], Gs[1]), psgd.precond_grad_kron(Qls[0], qn, Gn),
                 psgd.precond_grad_kron(Qls[2], Qrs[2], Gs[2].bfloat16()), psgd.precond_grad_kron(Qls[3], Qrs[3], Gs[3])]
    for got, want in zip(mixed, (want_a[0], ref_big, want_a[1], ref_nd, ref_bf, want_a[3])):
        assert torch.equal(got, want)
    with pytest.raises(RuntimeError):
        with kron.layer_batch():
            with kron.layer_streams():
                pass
    # an exception inside the block leaves the module usable (nothing is launched for the abandoned queue)
    with pytest.raises(ZeroDivisionError):
        with kron.layer_batch():
            dropped = psgd.precond_grad_kron(Qls[0], Qrs[0], Gs[0])
            dropped_u = psgd.update_precond_kron(Qls[1], Qrs[1], Gs[1], Gs[1], 0.01) if False else None
            1 / 0
    assert torch.isnan(dropped).all()                     # (what the abandoned call had handed out is marked, not stale memory)
    assert torch.equal(psgd.precond_grad_kron(Qls[0], Qrs[0], Gs[0]), want_a[0])
    # captured: the graph holds the batched launches
    n5 = 5
    outs = [torch.empty_like(x) for x in Gs[:n5]]

    def fn():
        with kron.layer_batch():
            res = [psgd.precond_grad_kron(a, b, g) for a, b, g in zip(Qls[:n5], Qrs[:n5], Gs[:n5])]
        for o, r_ in zip(outs, res):
            o.copy_(r_)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for o in outs:
        o.zero_()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    for o, c in zip(outs, want_a[:n5]):
        assert torch.equal(o, c)


@pytest.mark.parametrize("M,N", [(4096, 4096), (2048, 1536), (1100, 2304), (257, 120)])
def test_reference_apply_route_is_reproducible(psgd, M, N):
    """The DEFAULT route (round 6; no route call here): always the association order of psgd.py:189-192 (Gram of the smaller side),
    no first- / second-sight switch: three consecutive calls with identical inputs return identical bits (the opt-in "auto"
    route's three calls take three paths -- direct, prepare + apply, prepared -- whose bits differ in the last places)."""
    from psgd_tf_amd import kron
    rng = np.random.default_rng(M + 5 * N)
    Ql, Qr = _dev((_tri_factor(rng, M) * 1.3).astype(np.float32)), _dev(_tri_factor(rng, N).astype(np.float32))
    G = _dev(rng.standard_normal((M, N)).astype(np.float32))
    key = (G.get_device(), M, N, kron._raw_stream(G.get_device()))
    assert kron._apply_route == "reference"
    try:
        kron.invalidate_factor_cache()
        outs, paths = [], []
        for _ in range(3):
            outs.append(psgd.precond_grad_kron(Ql, Qr, G))
            paths.append(kron._apply_slots[key].path)
        assert paths == ["both", "prepared", "prepared"], paths
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
        fresh = psgd.precond_grad_kron(Ql.clone(), Qr.clone(), G)          # new tensor objects, same values: same bits again
        assert kron._apply_slots[key].path == "both" and torch.equal(fresh, outs[0])
        if min(M, N) <= 2304:
            ref = orc.precond_grad_kron(*(t.cpu().numpy().astype(np.float64) for t in (Ql, Qr, G)))
            assert rel_err(outs[0].cpu().numpy(), ref) < TOL
    finally:
        pass
    with pytest.raises(ValueError):
        kron.set_apply_route("fastest")
    if kron._apply_slots[key].fn_direct is not None:                       # large layer: the opt-in route starts on the direct chain
        assert kron.set_apply_route("auto") == "reference"
        try:
            kron.invalidate_factor_cache()
            psgd.precond_grad_kron(Ql, Qr, G)
            assert kron._apply_slots[key].path == "direct"
        finally:
            assert kron.set_apply_route("reference") == "auto"


@pytest.mark.parametrize("M,N", [(1024, 1024), (1100, 520), (640, 2304), (2048, 1536)])
def test_large_apply_paths_direct_both_prepared(psgd, M, N):
    """Large fp32 layers: factors seen for the first time take the Gram-free chain Ql' (Ql ((G Qr') Qr)) (psgd_kron_dd_apply_direct_f32:
    nothing prepared), the same factor tensors a second time make their Gram (prepare + apply), from then on the prepared half runs.
    All three agree with the fp64 oracle to the parity tolerance; an in-place change of a factor starts over."""
    from psgd_tf_amd import kron
    rng = np.random.default_rng(M + 3 * N)
    Ql, Qr = _dev((_tri_factor(rng, M) * 1.3).astype(np.float32)), _dev(_tri_factor(rng, N).astype(np.float32))
    Gs = [_dev(rng.standard_normal((M, N)).astype(np.float32)) for _ in range(4)]
    key = (Gs[0].get_device(), M, N, kron._raw_stream(Gs[0].get_device()))
    paths = []
    old = kron.set_apply_route("auto")                     # (the opt-in fast path; the default is the reproducible route)
    try:
        for G in Gs:
            out = psgd.precond_grad_kron(Ql, Qr, G)
            ref = orc.precond_grad_kron(*(t.cpu().numpy().astype(np.float64) for t in (Ql, Qr, G)))
            assert rel_err(out.cpu().numpy(), ref) < TOL
            paths.append(kron._apply_slots[key].path)
        assert paths == ["direct", "both", "prepared", "prepared"], paths
        Ql.mul_(1.01)                                          # (version counter moves: new factors)
        out = psgd.precond_grad_kron(Ql, Qr, Gs[0])
        assert kron._apply_slots[key].path == "direct"
        ref = orc.precond_grad_kron(*(t.cpu().numpy().astype(np.float64) for t in (Ql, Qr, Gs[0])))
        assert rel_err(out.cpu().numpy(), ref) < TOL
    finally:
        kron.set_apply_route(old)


def test_triangular_contract_check_is_opt_in(psgd):
    """psgd.py:173, :179, :190, :192 multiply with the full factors; the kernels ASSUME upper-triangular factors (triangular K ranges skip
    whole tiles below the diagonal, small layers and diagonal tiles multiply what is there): a factor with entries below its diagonal
    gives a result that is neither the reference's nor its upper triangle's -- silently by default, and with
    kron.set_triangular_check(True) the call raises instead."""
    from psgd_tf_amd import kron
    rng = np.random.default_rng(3)
    M, N = 40, 24
    Ql, Qr = _tri_factor(rng, M).astype(np.float32), _tri_factor(rng, N).astype(np.float32)
    G = rng.standard_normal((M, N)).astype(np.float32)
    bad = Ql.copy()
    bad[7, 2] = 0.5
    want = psgd.precond_grad_kron(_dev(Ql), _dev(Qr), _dev(G))
    assert torch.isfinite(psgd.precond_grad_kron(_dev(bad), _dev(Qr), _dev(G))).all()       # (no check by default: whatever it is, no error)
    old = kron.set_triangular_check(True)
    try:
        assert old is False
        assert torch.equal(psgd.precond_grad_kron(_dev(Ql), _dev(Qr), _dev(G)), want)
        with pytest.raises(ValueError):
            psgd.precond_grad_kron(_dev(bad), _dev(Qr), _dev(G))
        with pytest.raises(ValueError):
            psgd.update_precond_kron(_dev(Ql), _dev(Qr + np.tril(Qr.T, -1)), _dev(G), _dev(G), 0.01)
        q1 = _dev(np.ones((1, M), np.float32))                                            # sparse formats are not square factors
        psgd.precond_grad_kron(q1, _dev(Qr), _dev(rng.standard_normal((M, N)).astype(np.float32)))
    finally:
        kron.set_triangular_check(False)
