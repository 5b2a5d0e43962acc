# What the driver runs at round end, in one go: GPU tests, smoke, default bench.
set -x
mkdir -p gpurun_out/check
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/check/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/check/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
python bench.py > gpurun_out/check/bench.json 2> gpurun_out/check/bench.err; tail -c 1500 gpurun_out/check/bench.json
