"""GPU: bench.py's contract, run small.  N = 1 as the driver runs it, and the N = 2 code path (process group, barrier +
synchronize bracket, MAX over ranks, one JSON line from rank 0) in its test mode with both ranks on the one GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_single_gpu_line(hip_lib):
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--rows", "4000000",
                        "--no-kron", "--cpu-budget-s", "2"], cwd=ROOT, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["metric"] == "uvd_update_apply_params_per_sec" and d["unit"] == "params/s" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 4000000 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])


@pytest.mark.timeout(600)
def test_bench_two_ranks_code_path(hip_lib):
    env = dict(os.environ, PSGD_BENCH_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29523", "bench.py", "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--rows", "4000000"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["rows_global"] == 8000000
    assert "TEST MODE" in d["config"]["parallelism"]
    assert abs(d["value"] - 8000000 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
