"""GPU: bench.py's contract, run small.  N = 1 as the driver runs it, and the N = 2 code path (self-launched ranks,
process group, barrier + synchronize bracket, MAX over ranks, one JSON line from rank 0) in its test mode with both
ranks on the one GPU."""
import json
import os
import subprocess
import sys

import pytest
import torch

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
                        "--no-kron", "--cpu-budget-s", "2", "--wide-rows", "2000000"], cwd=ROOT, capture_output=True, text=True,
                       timeout=580)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    # the ONE line stays under the 8 KB the driver's record keeps; the prose lives in DESIGN.md, the full record on stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][0]
    assert len(line) < 8000 and '"note"' not in line
    full = [l for l in r.stderr.splitlines() if l.startswith("BENCH_DETAIL ")]
    assert len(full) == 1 and json.loads(full[0][13:])["value"] == pytest.approx(d["value"], rel=1e-5)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] is None                                  # one GPU: neither weak nor strong
    pl = d["config"]["placement"]                                # the state's allocator ran its probe and says what it kept
    assert pl["mode"] == "probe" and pl["layout"] and pl["candidates"] >= 1 and pl["held_gib"] > 0 and "log" not in pl
    # the figures the north star names, as scalars of `roofline` (what the driver's record keeps)
    rf = d["roofline"]
    for k in ("apply_ms", "apply_frac", "apply_frac_moved", "update_ms", "update_frac", "step_frac", "step_frac_moved",
              "step_two_reference_calls_ms", "config2_step_us", "uvd_r64_update_x_spec", "splu_r40_update_x_spec"):
        assert isinstance(rf[k], float) and rf[k] > 0, k
    assert rf["apply_frac"] == pytest.approx(340 * 4000000 / (rf["apply_ms"] * 1e-3) / 8e12, rel=1e-3)
    cb = d["cpu_baseline"]
    assert 1 <= cb["cores"] <= cb["thread_sweep_max"] <= cb["logical_cpus"]          # (the default 15-s budget sweeps to every CPU)
    assert cb["full_n_value"] is None or cb["full_n_value"] > 1e6
    assert d["metric"] == "uvd_update_apply_params_per_sec" and d["unit"] == "params/s" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 4000000 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"} <= set(d["roofline"])
    ex = d["exchange_overhead"]             # one rank's share of configs[3] unsharded vs through the 1-rank RCCL path
    assert "error" not in ex, ex
    assert ex["rows"] == 12500032 and ex["unsharded_ms"] > 0 and ex["sharded_1rank_rccl_ms"] > 0 and ex["backend"] == "nccl"
    two = d["roofline"]["paths"]["step_two_reference_calls"]
    assert two["wall_ms"] > 0 and two["params_per_s"] > 0
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    # the legs the north star names: apply alone and update alone with their own kernel times and non-null fractions,
    # the fused step on bytes moved as well as on SURVEY's algorithmic bytes, and config 2 (N = 1M, r = 10)
    paths = d["roofline"]["paths"]
    for leg, kernels in (("apply", ("apply_s1", "apply_s2", "apply_s3")), ("update", ("update_s1", "update_s2", "update_s3"))):
        assert paths[leg]["frac"] > 0 and paths[leg]["frac_moved"] > 0 and paths[leg]["wall_ms"] >= 0.9 * paths[leg]["kernel_ms"]
        assert all(paths[leg]["kernels_ms"][k] > 0 for k in kernels)
    assert paths["apply"]["alg_bytes_per_param"] == 340 and paths["apply"]["moved_bytes_per_param"] == 272
    assert paths["update"]["alg_bytes_per_param"] == 440
    assert paths["step"]["alg_bytes_per_param"] == 780 and paths["step"]["moved_bytes_per_param"] == 612
    assert 0 < paths["step"]["frac_moved"] < paths["step"]["frac"]
    c2 = d["config2_N1M_r10"]
    assert c2["N"] == 1_000_000 and c2["r"] == 10 and c2["step_fused"]["wall_ms"] > 0 and c2["apply"]["wall_ms"] > 0


@pytest.mark.timeout(600)
def test_bench_two_ranks_self_launched(hip_lib):
    """`python bench.py --gpus 2` from a plain shell (as the driver's scaling run starts it): the parent launches the two
    ranks itself and relays rank 0's line.  Test mode: both ranks on the one GPU, gloo collectives."""
    env = dict(os.environ, PSGD_BENCH_SINGLE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "4000000"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["rows_global"] == 8000000
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["collective_backend"] == "gloo"
    assert "TEST MODE" in d["config"]["parallelism"]
    assert abs(d["value"] - 8000000 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


@pytest.mark.timeout(900)
def test_bench_two_ranks_default_is_baseline_config3(hip_lib):
    """`python bench.py --gpus 2` with no row flags = BASELINE configs[3]: N_global = 100M rows, r = 20, split over the
    ranks in contiguous row blocks (strong scaling), plus the 100M-rows-per-GPU weak sub-record.  On a box with one GPU:
    test mode (both ranks on it, gloo) -- checks the workload, not the speed.  With two or more GPUs visible: the real thing,
    rank k on cuda:k over RCCL."""
    two = torch.cuda.device_count() >= 2
    env = dict(os.environ)
    env.pop("PSGD_BENCH_SINGLE_DEVICE", None)
    if not two:
        env["PSGD_BENCH_SINGLE_DEVICE"] = "1"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=880)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["collective_backend"] == ("nccl" if two else "gloo")
    assert ("TEST MODE" in d["config"]["parallelism"]) == (not two)
    assert d["config"]["rows_global"] == 100_000_000 and d["config"]["rows_per_gpu"] == 50_000_000
    assert d["config"]["rank_of_modification"] == 20 and d["config"]["baseline_config"].startswith("configs[3]")
    assert abs(d["value"] - 100_000_000 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    w = d["weak"]
    assert w["scaling"] == "weak" and w["rows_per_gpu"] == 100_000_000 and w["rows_global"] == 200_000_000
    assert abs(w["value"] - 200_000_000 * 2 / (w["ms_per_step"] * 2e-3)) < 1e-6 * w["value"]


@pytest.mark.timeout(600)
def test_bench_two_ranks_strong_ragged(hip_lib):
    """--global-rows that does not divide: rank blocks start at multiples of 64 rows, the last rank holds the rest."""
    env = dict(os.environ, PSGD_BENCH_SINGLE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--global-rows", "5000001"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["scaling"] == "strong" and d["config"]["rows_global"] == 5000001 and d["config"]["rows_per_gpu"] == 2500032
    assert "weak" not in d
    assert abs(d["value"] - 5000001 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


@pytest.mark.timeout(600)
def test_bench_two_ranks_under_torchrun(hip_lib):
    """The same job started the other way the contract allows: as ranks of torch.distributed.run."""
    env = dict(os.environ, PSGD_BENCH_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29523", "bench.py", "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--rows", "4000000"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["rows_global"] == 8000000


@pytest.mark.timeout(600)
def test_bench_one_rank_rccl_group(hip_lib):
    """--force-sharded at world size 1: the multi-GPU code path (process group on RCCL, all-gathers, fold kernels)."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--rows", "4000000",
                        "--force-sharded", "--no-kron", "--no-legs", "--no-cpu-baseline"], cwd=ROOT, capture_output=True,
                       text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["config"]["collective_backend"] == "nccl" and d["config"]["rccl_ranks"] == 1 and d["value"] > 0
