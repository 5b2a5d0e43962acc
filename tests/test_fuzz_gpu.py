"""A short randomised parity sweep (tests/fuzz_gpu.py) as part of the GPU suite: random shapes for the Kron fp32 and
bf16 paths, the fused UVd step and the sparse LU against fp64 torch restatements; tolerances as in the fixed-shape tests."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_stay_within_tolerance(hip_lib):
    from tests import fuzz_gpu
    cases, bad, worst = fuzz_gpu.run(12.0, seed=20240)
    assert cases > 100          # (a cold box spends seconds of the budget on the first calls)
    assert not bad, bad[:5]
